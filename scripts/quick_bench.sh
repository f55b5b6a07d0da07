#!/bin/bash
# GPU box: gpu tests, then bench variants (history/summary/full_trace x ieee/fast). Output: gpurun_out/
set -u
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
rm -f $OUT/variants.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for v in "--policy ieee" "--policy fast" "--mode summary --policy ieee" "--mode summary --policy fast" "--mode full_trace --policy ieee" "--mode full_trace --policy fast"; do
  echo "### bench.py $v" >> $OUT/variants.log
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v >> $OUT/variants.log 2>&1 || exit 1
done
python - <<'PY'
import json
cur=None
for line in open('/root/repo/gpurun_out/variants.log'):
    if line.startswith('###'): cur=line.strip()
    elif line.startswith('{'):
        j=json.loads(line)
        print(f"{cur:45s} value={j['value']:.3e} kernel_ms={j['roofline']['kernel_ms']:.3f} GB/s={j['roofline']['achieved']:.0f} frac={j['roofline']['frac']:.3f}")
PY
