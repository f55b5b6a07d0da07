#!/bin/bash
# GPU box: interleaved A/B of engine builds (build/variants/libort_*.so) on config 3 (statistics-only and full routes)
# and config 2 summary; prints ms per call per variant and round.   bash scripts/ab_config3.sh [rounds]
cd /root/repo
OUT=/root/repo/gpurun_out/ab_config3.log
rm -f $OUT
for round in $(seq 1 ${1:-3}); do
  for lib in $(ls build/variants/libort_*.so); do
    for wm in "config3 stats" "config3 full" "config2 summary"; do
      set -- $wm
      r=$(timeout -k 10 120 python scripts/run_workload.py $1 --mode $2 --reps 20 --lib $lib 2>&1 | tail -1)
      echo "$(basename $lib) $1 $2 round $round $(echo $r | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms" % d["ms"], d.get("survivors",""), d.get("mean_rms",""))' 2>/dev/null || echo FAILED $r)" >> $OUT
    done
  done
done
cat $OUT
