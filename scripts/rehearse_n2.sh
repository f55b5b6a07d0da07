#!/bin/bash
# GPU box (1 GPU): rehearsal of the N = 2 launch path of bench.py under an external launcher (the driver's form) — the
# weak-scaling headline, then BASELINE config 4 sharded over the ranks with the all-gather inside that leg's timed region —
# with the gloo backend (both ranks on cuda:0, the collective on CPU tensors), the exchange leg at reduced size, including
# rank 0's bit-for-bit check of the gathered hits against its own single-rank trace.  (`python bench.py --gpus 2` launches
# the ranks itself: scripts/final_profile.sh rehearses that form.)
cd /root/repo
export ORT_BENCH_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --zoom 4 --pupil4 256 > gpurun_out/rehearse_n2.log 2>&1
echo "rc=$?" >> gpurun_out/rehearse_n2.log
tail -c 2500 gpurun_out/rehearse_n2.log
