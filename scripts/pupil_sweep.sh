#!/bin/bash
# GPU box: bench kernel vs pupil size (history mode, fast policy): where the launch stops filling the chip
cd /root/repo
for k in 64 128 256 512 1024 2048; do
  python bench.py --pupil $k --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | K=$k python -c "
import sys, json, os
j = json.loads(sys.stdin.read()); r = j['roofline']
print('pupil %4s  rays %9d  kernel %.4f ms  %.3e intersections/s  %5.0f GB/s  frac %.3f' % (os.environ['K'], j['config']['rays_per_step_per_gpu'], r['kernel_ms'], j['value'], r['achieved'], r['frac']))"
done
