#!/bin/bash
# GPU box: A/B of engine builds (build/variants/libort_*.so) on the finite-conjugate launch rule (config 2 / config 3 bundles,
# summary / history / statistics-only), interleaved.   bash scripts/ab_raybasis.sh
cd /root/repo
OUT=/root/repo/gpurun_out/ab_raybasis.log
rm -f $OUT
for round in 1 2 3; do
  for lib in $(ls build/variants/libort_*.so); do
    for wm in "config2 summary" "config2 history" "config3 stats"; do
      set -- $wm
      r=$(timeout -k 10 120 python scripts/run_workload.py $1 --mode $2 --reps 20 --raybasis --lib $lib 2>&1 | tail -1)
      echo "$(basename $lib) $1 $2 raybasis round $round $(echo $r | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms" % d["ms"], d.get("survivors",""), d.get("mean_rms",""))' 2>/dev/null || echo FAILED $r)" >> $OUT
    done
  done
done
cat $OUT
