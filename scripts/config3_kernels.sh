#!/bin/bash
# GPU box: config 3 through full_trace (vectors out) under rocprofv3 --kernel-trace --stats, with the mixed-traffic
# ceiling of the placement pass (tools/place_ceiling) and the store ceiling measured in the same call (same box).
# bash scripts/config3_kernels.sh [tag]  -> gpurun_out/<tag>.log, gpurun_out/<tag>_stats/
TAG=${1:-c3_kernels}
OUT=/root/repo/gpurun_out
timeout -k 10 60 /root/repo/tools/place_ceiling > $OUT/$TAG.log 2>&1 || exit 1
timeout -k 10 100 /root/repo/tools/store_ceiling --quick >> $OUT/$TAG.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/${TAG}_stats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 /root/repo/scripts/run_workload.py config3 --mode full --reps 10 > $OUT/${TAG}_run.log 2>&1 || exit 1
tail -1 $OUT/${TAG}_run.log >> $OUT/$TAG.log
python3 - $OUT/${TAG}_stats >> $OUT/$TAG.log <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "ort::" in r["Name"] and float(r["AverageNs"]) > 5000:
        print(r["Name"][:80], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
cat $OUT/$TAG.log
