#!/bin/bash
# GPU box: effective clock (GRBM_GUI_ACTIVE / 8 / kernel time) + VALU busy of the fast kernel, history vs summary.
set -u
OUT=/root/repo/gpurun_out/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in history summary; do
  timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/$mode -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --policy fast --mode $mode > $OUT/$mode.log 2>&1 || { tail -5 $OUT/$mode.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for mode in ("history","summary"):
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob(f"/root/repo/gpurun_out/clock/{mode}/**/*counter_collection.csv", recursive=True):   # (bench.py's helper child writes one too)
        for r in csv.DictReader(open(f)):
            if "k_trace<double, 1" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9)
    # drop the other-policy launches (they are a different kernel name) and warmups: use medians
    med=lambda v: sorted(v)[len(v)//2]
    t=med(dur); g=med(acc["GRBM_GUI_ACTIVE"])
    print(mode, "kernel_ms %.3f"%(t*1e3), "clock_GHz %.3f"%(g/8/t/1e9), "VALU_insts/ray-surface %.1f"%(med(acc["SQ_INSTS_VALU"])*64/113246208/1),
          "SALU/wave %.0f"%(med(acc["SQ_INSTS_SALU"])/73728), "valu_active_quad/sq_busy %.3f"%(med(acc["SQ_ACTIVE_INST_VALU"])/med(acc["SQ_BUSY_CYCLES"])),
          "wave_cycles %.3e"%med(acc["SQ_WAVE_CYCLES"]), "wait_inst_any %.3e"%med(acc["SQ_WAIT_INST_ANY"]), "wait_any %.3e"%med(acc["SQ_WAIT_ANY"]), "valu_active %.3e"%med(acc["SQ_ACTIVE_INST_VALU"]))
PY
