#!/bin/bash
# GPU box: SQ counters (three passes of <= 8 counters, counters + kernel trace only) of every ort:: kernel of ONE
# workload run:   bash scripts/pmc_sq.sh <tag> <scripts/run_workload.py arguments...>
# -> gpurun_out/<tag>/summary.json  {kernel: {counter: mean per dispatch}}
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"
P3="SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 /root/repo/scripts/run_workload.py "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ort::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: sum(v) / len(v) for c, v in d.items()}, _dispatches=max(len(v) for v in d.values())) for k, d in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, d in res.items():
    if d.get("SQ_INSTS_VALU", 0) > 1e6:
        print(k[:90], {c: round(v) for c, v in sorted(d.items())})
PY
