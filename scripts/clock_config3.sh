#!/bin/bash
# GPU box: what bounds the config-3 trace kernel — effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time), VALU issue
# utilisation and instruction counts, for one or more engine builds:  bash scripts/clock_config3.sh <tag> [lib ...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIBS="${@:-default}"
for lib in $LIBS; do
  n=$(basename $lib .so)
  if [ "$lib" = default ]; then L=""; else L="--lib $lib"; fi
  timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_WAVES --kernel-trace --output-format csv -d $OUT/$n -- python3 /root/repo/scripts/run_workload.py config3 --mode stats --reps 6 $L > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, os, json
out = sys.argv[1]; res = {}
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f[0])):
        if "k_trace<double, 1, 2" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    med = lambda v: sorted(v)[len(v) // 2]
    t = med(dur); m = {k: med(v) for k, v in acc.items()}
    res[os.path.basename(d.rstrip("/"))] = dict(kernel_ms=t * 1e3, clock_GHz=m["GRBM_GUI_ACTIVE"] / 8 / t / 1e9,
        valu_per_wave=m["SQ_INSTS_VALU"] / m["SQ_WAVES"], trans_per_wave=m.get("SQ_INSTS_VALU_TRANS_F64", 0) / m["SQ_WAVES"],
        valu_issue_utilisation=m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CU_CYCLES"], active_valu_per_wave=m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVES"],
        wave_cycles_per_wave=m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"], wait_inst_any_per_wave=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVES"], **{k: v for k, v in m.items()})
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: (round(b, 4) if b < 1e4 else round(b)) for a, b in v.items()})
PY
