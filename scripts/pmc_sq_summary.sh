#!/bin/bash
# GPU box: SQ counters of the summary-mode trace kernel (config 2 shape), both policies — what its waves wait for.
# Counters only (no trace domains besides --kernel-trace), three passes of <= 8 counters.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/sq_summary
rm -rf $OUT; mkdir -p $OUT
Q="--steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 --no-ceiling --mode summary"
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"
P3="SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
for pol in fast ieee; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${pol}_$i -- python3 /root/repo/bench.py $Q --policy $pol > $OUT/${pol}_$i.log 2>&1 || { tail -5 $OUT/${pol}_$i.log; exit 1; }
  done
done
python3 - <<'PY'
import csv,glob,collections,json
out={}
for pol in ("fast","ieee"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"/root/repo/gpurun_out/sq_summary/{pol}_*/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_trace<" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[pol]={k:sum(v)/len(v) for k,v in acc.items()}
    print(pol, {k:round(v) for k,v in sorted(out[pol].items())})
json.dump(out,open("/root/repo/gpurun_out/sq_summary/summary.json","w"),indent=1)
PY
