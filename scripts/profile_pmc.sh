#!/bin/bash
# GPU box: PMC passes (counters only + kernel trace, each in its own run) for the bench kernel.
# usage: profile_pmc.sh "<bench flags>" <tag>
set -u
FLAGS="${1:-}"
TAG="${2:-ieee}"
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline $FLAGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY || exit 1
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
echo done
