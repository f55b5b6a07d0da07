#!/bin/bash
# GPU box (1 GPU): rehearsal of the N = 2 launch path with the gloo backend (both ranks on cuda:0).
cd /root/repo
export ORT_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --pupil 512 > gpurun_out/rehearse_n2.log 2>&1
echo "rc=$?" >> gpurun_out/rehearse_n2.log
tail -5 gpurun_out/rehearse_n2.log
# config 4 driver, 2 ranks, gloo, reduced size, with the bit-for-bit check against the single-rank trace
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 scripts/zoom_sweep_dist.py --pupil 128 --zoom 8 --check > gpurun_out/rehearse_zoom_n2.log 2>&1
echo "rc=$?" >> gpurun_out/rehearse_zoom_n2.log
tail -6 gpurun_out/rehearse_zoom_n2.log
