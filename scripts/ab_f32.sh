#!/bin/bash
# GPU box: A/B of build variants on the Float32 statistics pipeline (config 5) and the Float32 list-mode history trace
cd /root/repo
OUT=/root/repo/gpurun_out
rm -f $OUT/ab_f32.log
for round in 1 2 3; do
  for lib in default $(ls /root/repo/build/variants/libort_*.so); do
    if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
    echo "### $(basename $lib) round $round" >> $OUT/ab_f32.log
    timeout -k 10 120 python scripts/spot_batch_timing.py 256 2>&1 | grep "f32 rep 3" >> $OUT/ab_f32.log || echo FAILED >> $OUT/ab_f32.log
    timeout -k 10 120 python scripts/list_mode.py 2>&1 | grep -i "f32" >> $OUT/ab_f32.log
  done
done
cat $OUT/ab_f32.log
