#!/bin/bash
# GPU box, round 2 step 3: gpu tests, FAST attribution (two seeds), A/B against the round-1 library (fast + ieee, history /
# summary / full_trace), config 3 full size both libraries, IEEE PMC.
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout -k 10 600 python scripts/fast_attribution.py 2024 300 > $OUT/fast_attr.log 2>&1; tail -45 $OUT/fast_attr.log
timeout -k 10 600 python scripts/fast_attribution.py 777 300 > $OUT/fast_attr2.log 2>&1; tail -45 $OUT/fast_attr2.log
rm -f $OUT/ab2.log
for round in 1 2 3; do
  for lib in default /root/repo/build/variants/libort_r1.so; do
    for v in "--policy fast --mode history" "--policy fast --mode summary" "--policy fast --mode full_trace" "--policy ieee --mode history" "--policy ieee --mode summary" "--policy ieee --mode full_trace"; do
      if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
      echo "### $(basename $lib) $v round $round" >> $OUT/ab2.log
      timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $v >> $OUT/ab2.log 2>&1 || echo FAILED >> $OUT/ab2.log
    done
  done
done
unset ORT_HIP_LIB
python - <<'PY'
import json, collections
cur=None; res=collections.defaultdict(list)
for line in open('/root/repo/gpurun_out/ab2.log'):
    if line.startswith('###'): cur=' '.join(line.split()[1:6])
    elif line.startswith('{'):
        j=json.loads(line); res[cur].append(j['roofline']['kernel_ms'])
    elif 'FAILED' in line: res[cur].append(float('nan'))
for k,v in sorted(res.items()):
    print(f"{k:55s} " + ' '.join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
PY
echo "### config3 new" > $OUT/config3_ab.log; timeout -k 10 300 python scripts/config3_demo.py >> $OUT/config3_ab.log 2>&1
echo "### config3 r1" >> $OUT/config3_ab.log; ORT_HIP_LIB=/root/repo/build/variants/libort_r1.so timeout -k 10 300 python scripts/config3_demo.py >> $OUT/config3_ab.log 2>&1
cat $OUT/config3_ab.log
bash scripts/profile_pmc.sh "--policy ieee" ieee_r2 2>&1 | tail -3
python scripts/pmc_summary.py ieee_r2 > $OUT/pmc_ieee_r2.json 2>&1; grep -E "SQ_INSTS_VALU|k_trace|GRBM" $OUT/pmc_ieee_r2.json
