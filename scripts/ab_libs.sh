#!/bin/bash
# GPU box: interleaved A/B of engine builds on one workload of scripts/run_workload.py.
#   bash scripts/ab_libs.sh <out.log> <rounds> "<run_workload arguments>" lib [lib ...]      (lib = default | path of a build)
OUT=/root/repo/gpurun_out/$1; ROUNDS=$2; ARGS=$3; shift 3
cd /root/repo
echo "## run_workload.py $ARGS" >> $OUT
for round in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then L=""; else L="--lib $lib"; fi
    echo -n "round $round $(basename $lib .so): " >> $OUT
    timeout -k 10 200 python scripts/run_workload.py $ARGS $L 2>/dev/null | tail -1 >> $OUT || echo FAILED >> $OUT
  done
done
