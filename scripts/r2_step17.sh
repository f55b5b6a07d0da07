#!/bin/bash
# GPU box: 6 waves per SIMD (80 VGPRs) for the kernels without history stores, against the 5-wave build, same box, interleaved.
cd /root/repo
OUT=/root/repo/gpurun_out/step17
rm -rf $OUT; mkdir -p $OUT
pick='
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
e=j.get("extra",{})
print(sys.argv[1], "summary", round(e["config2_summary"]["kernel_ms"],4), "| c3 ft", "%.3e"%e["config3_full_trace"]["value"], "stats", "%.3e"%e["config3_statistics_only"]["value"], "| c5 f32", "%.3e"%e["config5_spot_batch_f32"]["value"])
'
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify --sustain-s 0 --no-ceiling --steps 5 --warmup 2 2>/dev/null | python -c "$pick" waves5 || exit 1
ORT_HIP_LIB=/root/repo/build/libort_w6.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify --sustain-s 0 --no-ceiling --steps 5 --warmup 2 2>/dev/null | python -c "$pick" waves6 || exit 1
done | tee $OUT/ab.log
for lib in "" /root/repo/build/libort_w6.so; do
  if [ -n "$lib" ]; then export ORT_HIP_LIB=$lib; else unset ORT_HIP_LIB; fi
  timeout -k 10 300 python bench.py --workload config4 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config4 lib=$lib', 'ms_per_step', round(j['ms_per_step'],3), 'trace only', round(j['gather_exclusive']['ms_per_step'],3), j['verified'])" | tee -a $OUT/ab.log
done
