#!/bin/bash
# GPU box: k_ft_place's average duration (rocprofv3 --kernel-trace --stats, 100 calls of config 3 full_trace) per engine build
#   bash scripts/place_kernel_ab.sh <tag> lib [lib ...]      (lib = default | path)
TAG=$1; shift
OUT=/root/repo/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so); if [ "$lib" = default ]; then L=""; else L="--lib /root/repo/$lib"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${n}_$round -- python3 /root/repo/scripts/run_workload.py config3 --mode full --reps 100 $L > $OUT/${n}_$round.log 2>&1
  f=$(ls -t $OUT/${n}_$round/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$n round $round" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ft_place" in r["Name"] or "k_trace<double, 1, 2" in r["Name"]:
        print(sys.argv[2], r["Name"][:48], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done; done
