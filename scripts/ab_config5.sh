#!/bin/bash
# GPU box: A/B of engine builds (build/variants/libort_*.so) on BASELINE config 5 (ort_spot_batch_f32, 10^4 instances):
# wall time per call and Float32-vs-Float64 hit accuracy.   bash scripts/ab_config5.sh
cd /root/repo
OUT=/root/repo/gpurun_out/ab_config5.log
rm -f $OUT
for lib in $(ls build/variants/libort_*.so); do
  echo "## $(basename $lib) accuracy" >> $OUT
  timeout -k 10 200 python scripts/f32_accuracy.py --lib $lib 2>&1 | grep -v amdgpu.ids >> $OUT || echo FAILED >> $OUT
done
for round in 1 2 3; do
  for lib in $(ls build/variants/libort_*.so); do
    echo "## $(basename $lib) round $round" >> $OUT
    ORT_HIP_LIB=/root/repo/$lib timeout -k 10 200 python scripts/spot_batch_timing.py 256 2>&1 | grep "spot_batch_f32 rep [23]" >> $OUT || echo FAILED >> $OUT
  done
done
cat $OUT
