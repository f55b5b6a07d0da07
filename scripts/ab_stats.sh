#!/bin/bash
# GPU box: A/B of compile-time variants on the statistics-only spot pipeline (config 5, 256x128 rays per bundle)
cd /root/repo
OUT=/root/repo/gpurun_out
rm -f $OUT/ab_stats.log
for round in 1 2 3; do
  for lib in default $(ls /root/repo/build/variants/libort_*.so); do
    if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
    echo "### $(basename $lib) round $round" >> $OUT/ab_stats.log
    timeout -k 10 120 python scripts/spot_batch_timing.py 256 2>&1 | grep "rep 3" >> $OUT/ab_stats.log || echo FAILED >> $OUT/ab_stats.log
  done
done
cat $OUT/ab_stats.log
