#!/bin/bash
# GPU box: the fused full_trace route (ORT_FT_FUSED) beside the default one on BASELINE config 3 — kernel times, effective clock
# (GRBM_GUI_ACTIVE / 8 XCDs / kernel time) and VALU issue utilisation per kernel, then board power sampled while each route loops.
#   bash scripts/clock_fused.sh <tag>
TAG=${1:-r4_clock_fused}
OUT=/root/repo/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in full fused stats; do
  timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/$m -- python3 /root/repo/scripts/run_workload.py config3 --mode $m --reps 6 > $OUT/$m.log 2>&1 || { tail -5 $OUT/$m.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, os, json
out = sys.argv[1]; res = {}
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "k_trace" not in k and "k_ft_" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    med = lambda v: sorted(v)[len(v) // 2]
    mode = os.path.basename(d.rstrip("/")); res[mode] = {}
    for k in acc:
        t = med(dur[k]); m = {c: med(v) for c, v in acc[k].items()}
        res[mode][k] = dict(kernel_ms=round(t * 1e3, 4), clock_GHz=round(m["GRBM_GUI_ACTIVE"] / 8 / t / 1e9, 3),
                            valu_issue_utilisation=round(m["SQ_ACTIVE_INST_VALU"] / max(1.0, m["SQ_BUSY_CU_CYCLES"]), 3),
                            insts_valu=m["SQ_INSTS_VALU"], waves=m["SQ_WAVES"])
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for mode, ks in res.items():
    for k, v in ks.items(): print(mode, k, v)
PY
# board power while each route loops (~6 s each), sampled once a second
for m in full fused stats; do
  python3 /root/repo/scripts/run_workload.py config3 --mode $m --reps 3000 > $OUT/loop_$m.log 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i -E "power|sclk" | head -4 | sed "s/^/$m: /" >> $OUT/power.log; sleep 1; done
  wait $pid
  grep '"ms"' $OUT/loop_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'loop ms', round(d['ms'],4))" >> $OUT/power.log
done
cat $OUT/power.log
# clock the sustained loops ran at, if a call's kernels take the same cycles there as under the profiler (issue-bound kernels do):
# sum over the call's kernels of (profiled time x profiled clock) / sustained time per call
python3 - $OUT <<'PY'
import json, re, sys
out = sys.argv[1]
res = json.load(open(out + "/summary.json"))
loops = dict(re.findall(r"^(\w+) loop ms ([\d.]+)", open(out + "/power.log").read(), re.M))
for mode, ks in res.items():
    cyc = sum(v["kernel_ms"] * 1e-3 * v["clock_GHz"] * 1e9 for k, v in ks.items() if "paraxial" not in k)
    prof_ms = sum(v["kernel_ms"] for k, v in ks.items() if "paraxial" not in k)
    if mode in loops:
        print(f"{mode}: kernels under the profiler {prof_ms:.4f} ms ({cyc / 1e6:.3f} M cycles per XCD); sustained loop {float(loops[mode]):.4f} ms per call"
              f" -> {cyc / (float(loops[mode]) * 1e-3) / 1e9:.3f} GHz if the cycles are the same")
PY
