#!/bin/bash
# GPU box, round 2 step 10: look-back full_trace route with 4 waves per SIMD (no spill) against the 5-wave build and the default
# placement route, same box; then the WRITE_SIZE / FETCH_SIZE passes of the look-back route.
cd /root/repo
OUT=/root/repo/gpurun_out/step10
rm -rf $OUT; mkdir -p $OUT
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 --no-ceiling"
pick='import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], "ms_per_step", round(j["ms_per_step"],4), "kernel_ms", j["roofline"].get("kernel_ms"))'
for rep in 1 2; do
timeout -k 10 120 python bench.py $Q --mode full_trace 2>/dev/null | python -c "$pick" place || exit 1
timeout -k 10 120 python bench.py $Q --mode full_trace --ft-lookback 2>/dev/null | python -c "$pick" lookback_4waves || exit 1
ORT_HIP_LIB=/root/repo/build/libort_lb5.so timeout -k 10 120 python bench.py $Q --mode full_trace --ft-lookback 2>/dev/null | python -c "$pick" lookback_5waves || exit 1
done | tee $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
Q="--steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 --no-ceiling"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/ft_write_lookback -- python3 /root/repo/bench.py $Q --mode full_trace --ft-lookback > $OUT/w.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ft_fetch_lookback -- python3 /root/repo/bench.py $Q --mode full_trace --ft-lookback > $OUT/f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lookback -- python3 /root/repo/bench.py $Q --mode full_trace --ft-lookback > $OUT/s.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
for tag in ("ft_write_lookback","ft_fetch_lookback"):
    f=glob.glob(f"/root/repo/gpurun_out/step10/{tag}/**/*counter_collection.csv",recursive=True)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60],r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "k_trace" in k[0] or "k_ft" in k[0]: print(tag,k,len(v),sum(v)/len(v)*1024/9437184.0,"B/ray (raw KB->B)")
f=glob.glob("/root/repo/gpurun_out/step10/stats_lookback/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_trace" in r["Name"] or "k_ft" in r["Name"]: print(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e6)
PY
