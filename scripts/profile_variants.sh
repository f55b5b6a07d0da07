#!/bin/bash
# Runs on the GPU box: bench variants + rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
for v in "" "--no-lds" "--fast-math" "--fast-math --no-lds" "--mode summary" "--mode summary --fast-math" "--mode full_trace" "--mode full_trace --fast-math"; do
  echo "### bench.py $v" >> $OUT/variants.log
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v >> $OUT/variants.log 2>&1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ieee -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_ieee.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fast -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --fast-math > $OUT/prof_fast.log 2>&1 || exit 1
echo done
