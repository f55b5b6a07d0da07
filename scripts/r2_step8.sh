#!/bin/bash
# GPU box, round 2 step 8: store ceiling of the history layout (sustained), then the parity soak of both policies.
cd /root/repo
OUT=/root/repo/gpurun_out
timeout -k 10 120 ./tools/store_ceiling > $OUT/store_ceiling.log 2>&1; cat $OUT/store_ceiling.log
timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --no-verify --sustain-s 1.0 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench same box: kernel_ms', j['roofline']['kernel_ms'], 'sustained', j['sustained']['kernel_ms'], j['sustained']['frac'])" | tee -a $OUT/store_ceiling.log
timeout -k 10 900 python scripts/soak_parity.py 1500 4242 > $OUT/soak_r2.log 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak_r2.log
