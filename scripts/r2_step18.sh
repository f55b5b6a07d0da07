#!/bin/bash
# GPU box: the N > 1 bench path at world 1 (native RCCL) for config 4 and config 5, with the spot-statistics extra, and the 2-rank gloo rehearsal.
cd /root/repo
OUT=/root/repo/gpurun_out/step18
rm -rf $OUT; mkdir -p $OUT
for w in config4 config5; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 > $OUT/bench_${w}_world1.json 2> $OUT/bench_${w}_world1.err; rc=$?
  echo "$w rc=$rc"; [ $rc -eq 0 ] || { tail -5 $OUT/bench_${w}_world1.err; exit $rc; }
  python - $OUT/bench_${w}_world1.json <<'PY'
import sys,json
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(' ms_per_step', round(j['ms_per_step'],3), 'trace only', round(j['gather_exclusive']['ms_per_step'],3), 'verified', j['verified'])
print(' spot statistics', {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['extra']['spot_statistics'].items() if k!='what'})
PY
done
bash scripts/rehearse_n2.sh > $OUT/rehearse.log 2>&1; rc=$?; echo "rehearse rc=$rc"; tail -c 1500 $OUT/rehearse.log
exit $rc
