#!/bin/bash
# GPU box: what bounds the Float32 kernels of BASELINE config 5 — effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time), VALU
# issue utilisation, instruction counts and the wait share of the statistics kernel (ort_spot_batch_f32) and of the Float32
# summary kernel tracing the same number of rays, for one or more engine builds:
#   bash scripts/clock_config5.sh <tag> [lib ...]        -> gpurun_out/<tag>/summary.json
TAG=$1; shift
OUT=/root/repo/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIBS="${@:-default}"
for lib in $LIBS; do
  n=$(basename $lib .so)
  if [ "$lib" = default ]; then L=""; else L="--lib $lib"; fi
  for mode in stats hits; do
    timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/${n}_$mode -- python3 /root/repo/scripts/run_workload.py config5 --mode $mode --reps 4 $L > $OUT/${n}_$mode.log 2>&1 || { tail -5 $OUT/${n}_$mode.log; exit 1; }
  done
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, os, json
out = sys.argv[1]; res = {}
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: continue
    per = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "k_trace<float" not in k: continue
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    med = lambda v: sorted(v)[len(v) // 2]
    for k, acc in per.items():
        if max(dur[k]) < 1e-3: continue                     # the warm-up's small launches
        t = med(dur[k]); m = {c: med(v) for c, v in acc.items()}
        rays = 655360000.0
        res[os.path.basename(d.rstrip("/")) + " " + k.split("(")[0]] = dict(kernel_ms=t * 1e3, clock_GHz=m["GRBM_GUI_ACTIVE"] / 8 / t / 1e9,
            valu_per_intersection=m["SQ_INSTS_VALU"] * 64 / (rays * 12), valu_per_wave=m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
            lds_insts_per_wave=m["SQ_INSTS_LDS"] / m["SQ_WAVES"],
            valu_issue_utilisation=m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CU_CYCLES"], active_valu_per_wave=m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVES"],
            wave_cycles_per_wave=m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"], wait_inst_any_per_wave=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVES"],
            wait_share=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], intersections_per_s=rays * 12 / t, **m)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: (round(b, 4) if b < 1e4 else round(b)) for a, b in v.items()})
PY
