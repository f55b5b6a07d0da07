#!/bin/bash
# GPU box, round 2 step 9: gpu tests (config-5 tolerance), then the config-5 Float32 hit payload through the sharded bench path.
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload config5 --steps 5 --warmup 2 > $OUT/bench_config5_world1.json 2> $OUT/bench_config5_world1.err; rc=$?
echo "config5 bench rc=$rc"; tail -c 2500 $OUT/bench_config5_world1.json; tail -3 $OUT/bench_config5_world1.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload config4 --steps 10 --warmup 3 > $OUT/bench_config4_world1.json 2> $OUT/bench_config4_world1.err; rc=$?
echo "config4 bench rc=$rc"; tail -c 1500 $OUT/bench_config4_world1.json
exit $rc
