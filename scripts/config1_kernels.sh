#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the reference's own call (config 1: Cooke triplet, full_trace(system, H, 64))
# at H = 0 and H = 1 -> gpurun_out/c1prof_<H>/ ; prints per-kernel average / minimum durations and the wall time.
cd /tmp && export TMPDIR=/tmp
for H in 0.0 1.0; do
  rm -rf /root/repo/gpurun_out/c1prof_$H
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c1prof_$H -- python3 /root/repo/scripts/run_workload.py config1 --mode full --field $H --reps 300 > /root/repo/gpurun_out/c1prof_$H.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("/root/repo/gpurun_out/c1prof_$H/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print("H=$H", r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"])
PY
  grep wall /root/repo/gpurun_out/c1prof_$H.log
done
