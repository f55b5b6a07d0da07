#!/bin/bash
# GPU box: A/B of compile-time variants (gpurun_out/libort_*.so), interleaved rounds, fast policy.
cd /root/repo
OUT=/root/repo/gpurun_out
rm -f $OUT/ab.log
for round in 1 2 3; do
  for lib in default $(ls /root/repo/build/variants/libort_*.so); do
    for mode in history summary; do
      if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
      echo "### $(basename $lib) $mode round $round" >> $OUT/ab.log
      timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fast-math --mode $mode >> $OUT/ab.log 2>&1 || echo FAILED >> $OUT/ab.log
    done
  done
done
python - <<'PY'
import json, collections
cur=None; res=collections.defaultdict(list)
for line in open('/root/repo/gpurun_out/ab.log'):
    if line.startswith('###'): cur=' '.join(line.split()[1:3])
    elif line.startswith('{'):
        j=json.loads(line); res[cur].append(j['roofline']['kernel_ms'])
    elif 'FAILED' in line: res[cur].append(float('nan'))
for k,v in sorted(res.items()):
    print(f"{k:32s} " + ' '.join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
PY
