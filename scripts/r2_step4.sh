#!/bin/bash
# GPU box, round 2 step 4: gpu tests, bench N=1 default, config-4 path at world 1 (native RCCL 1 rank) and a 2-rank gloo
# rehearsal, kernel trace of the full_trace pipeline, A/B of FT vs r1.
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout -k 10 600 python bench.py > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"; tail -c 6000 $OUT/bench_n1.log
timeout -k 10 300 python bench.py --workload config4 --steps 5 --warmup 2 > $OUT/bench_c4_w1.log 2>&1; echo "c4 rc=$?"; tail -c 3000 $OUT/bench_c4_w1.log
ORT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --zoom 4 --pupil4 256 > $OUT/bench_c4_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -c 2500 $OUT/bench_c4_gloo2.log
rm -f $OUT/ab3.log
for round in 1 2 3; do
  for lib in default /root/repo/build/variants/libort_r1.so; do
    for v in "--policy fast --mode history" "--policy fast --mode summary" "--policy fast --mode full_trace" "--policy ieee --mode history"; do
      if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=$lib; fi
      echo "### $(basename $lib) $v round $round" >> $OUT/ab3.log
      timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 $v >> $OUT/ab3.log 2>&1 || echo FAILED >> $OUT/ab3.log
    done
  done
done
unset ORT_HIP_LIB
python - <<'PY'
import json, collections
cur=None; res=collections.defaultdict(list)
for line in open('/root/repo/gpurun_out/ab3.log'):
    if line.startswith('###'): cur=' '.join(line.split()[1:6])
    elif line.startswith('{'):
        j=json.loads(line); res[cur].append(j['roofline']['kernel_ms'])
    elif 'FAILED' in line: res[cur].append(float('nan'))
for k,v in sorted(res.items()):
    print(f"{k:55s} " + ' '.join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ft_r2 -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 --mode full_trace > $OUT/prof_ft_r2.log 2>&1
f=$(find $OUT/prof_ft_r2 -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f
