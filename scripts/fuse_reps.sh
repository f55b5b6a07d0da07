set -e
cd /root/repo
for reps in 8 40 200 1000; do
for m in full fused stats; do
timeout -k 10 120 python scripts/run_workload.py config3 --mode $m --reps $reps | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['mode'], d['reps'], round(d['ms'],4))
"
done; done
