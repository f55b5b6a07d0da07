#!/bin/bash
# GPU box, round 2 step 11: where the look-back trace kernel's extra time goes (no-spill builds): per-kernel times of
# the placement route, the look-back route, and the look-back route with the wait / the ticket / both removed (A/B builds).
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/step11
rm -rf $OUT; mkdir -p $OUT
Q="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 --no-ceiling --mode full_trace"
run() {   # tag lib extra
  if [ -n "$2" ]; then export ORT_HIP_LIB=$2; else unset ORT_HIP_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -- python3 /root/repo/bench.py $Q $3 > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; exit 1; }
  python3 - $1 <<'PY'
import csv,glob,sys
tag=sys.argv[1]
f=sorted(glob.glob(f"/root/repo/gpurun_out/step11/{tag}/**/*kernel_stats.csv",recursive=True))[-1]
tot=0; parts=[]
for r in csv.DictReader(open(f)):
    if "k_trace<" in r["Name"] or "k_ft" in r["Name"]:
        ms=float(r["AverageNs"])/1e6; tot+=ms; parts.append(f'{r["Name"].split("(")[0][-28:]}={ms:.4f}')
print(f"{tag:16s} total {tot:.4f} ms | "+" ".join(parts))
PY
}
for rep in 1 2; do
run place "" ""
run place_4waves /root/repo/build/libort_full4.so ""
run lookback "" "--ft-lookback"
run lb_nowait /root/repo/build/libort_dbg1.so "--ft-lookback"
run lb_noticket /root/repo/build/libort_dbg2.so "--ft-lookback"
run lb_neither /root/repo/build/libort_dbg3.so "--ft-lookback"
done 2>&1 | tee $OUT/summary.log
