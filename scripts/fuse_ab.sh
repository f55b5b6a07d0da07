set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_route" 2>&1 | tail -25
for r in 1 2 3; do
for m in full fused lookback; do
timeout -k 10 120 python scripts/run_workload.py config3 --mode $m --reps 8
done; done
