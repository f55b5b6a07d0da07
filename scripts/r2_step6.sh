#!/bin/bash
# GPU box, round 2 step 6: gpu tests; full_trace routes A/B (tile-local compaction + placement, look-back, round-1 dense);
# default bench line.
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
rm -f $OUT/ab6.log
for round in 1 2 3; do
  for v in "default:--mode full_trace" "default:--mode full_trace --ft-lookback" "r1:--mode full_trace" "default:--mode history" "r1:--mode history" "default:--mode history --policy ieee" "r1:--mode history --policy ieee"; do
    lib=${v%%:*}; a=${v#*:}
    if [ "$lib" = default ]; then unset ORT_HIP_LIB; else export ORT_HIP_LIB=/root/repo/build/variants/libort_$lib.so; fi
    echo "### $lib $a round $round" >> $OUT/ab6.log
    timeout -k 10 120 python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-extras --no-verify --sustain-s 0 $a >> $OUT/ab6.log 2>&1 || echo FAILED >> $OUT/ab6.log
  done
done
unset ORT_HIP_LIB
python - <<'PY'
import json, collections
cur=None; res=collections.defaultdict(list)
for line in open('/root/repo/gpurun_out/ab6.log'):
    if line.startswith('###'): cur=' '.join(line.split()[1:-2])
    elif line.startswith('{'):
        j=json.loads(line); res[cur].append(j['roofline']['kernel_ms'])
    elif 'FAILED' in line: res[cur].append(float('nan'))
for k,v in sorted(res.items()):
    print(f"{k:55s} " + ' '.join(f"{x:.3f}" for x in v) + f"   min {min(v):.3f}")
PY
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for line in open('/root/repo/gpurun_out/bench_n1.log'):
    if line.startswith('{'):
        j=json.loads(line)
        print("value %.4e ms/step %.4f frac %.3f | sustained ms %.4f frac %.3f | ieee ms %.4f frac %.3f sustained %.4f %.3f | verified %s" % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['sustained']['kernel_ms'], j['sustained']['frac'], j['other_policy']['kernel_ms'], j['other_policy']['frac'], j['other_policy']['sustained_kernel_ms'], j['other_policy']['sustained_frac'], j['verified']))
        for k,v in j['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('kernel_ms','pipeline_ms','value','frac')})
PY
