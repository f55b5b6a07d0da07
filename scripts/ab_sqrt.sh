#!/bin/bash
cd /root/repo
for lib in default /root/repo/build/variants/libort_sqrtq.so; do
  if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
  echo "### $lib"
  timeout -k 10 200 python scripts/fast_accuracy.py 2>&1 | grep config
done
bash scripts/ab_variants.sh
