#!/bin/bash
# GPU box: the judged artefacts — default bench line, rocprofv3 --kernel-trace --stats of the SAME
# command, and the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, counters only).
set -u
cd /root/repo
OUT=/root/repo/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 /root/repo/bench.py > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
for pol in fast ieee; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$pol -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --policy $pol > $OUT/fetch_$pol.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$pol -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --policy $pol > $OUT/write_$pol.log 2>&1 || exit 1
done
python3 /root/repo/scripts/collect_final.py
