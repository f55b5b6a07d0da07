#!/bin/bash
# GPU box: A/B of build variants (build/variants/libort_*.so vs the in-tree library) on both arithmetic policies
# (history mode) and on config 3 (aspheric rows), interleaved rounds.
cd /root/repo
OUT=/root/repo/gpurun_out
rm -f $OUT/ab_pol.log
for round in 1 2 3; do
  for lib in default $(ls /root/repo/build/variants/libort_*.so); do
    if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
    for pol in fast ieee; do
      echo "### $(basename $lib) $pol round $round" >> $OUT/ab_pol.log
      timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --policy $pol 2>&1 | python -c "import sys,json; [print(json.loads(l)['roofline']['kernel_ms']) for l in sys.stdin if l.startswith('{')]" >> $OUT/ab_pol.log || echo FAILED >> $OUT/ab_pol.log
    done
    echo "### $(basename $lib) config3 round $round" >> $OUT/ab_pol.log
    timeout -k 10 200 python scripts/config3_demo.py 1024 2>&1 | grep -E "history|statistics" >> $OUT/ab_pol.log
  done
done
cat $OUT/ab_pol.log
