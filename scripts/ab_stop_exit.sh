#!/bin/bash
# GPU box: the full_trace kernels with and without the early end of the surface loop for waves that lie entirely outside the stop
# (default: Float64 statistics kernels + polynomial builds; build/variants/libort_nse.so: -DORT_STOP_EXIT=0), sustained clocks
cd /root/repo
V="build/variants/libort_nse.so"
bash scripts/ab_libs.sh $1 2 "config3 --mode stats --reps 300" default $V
bash scripts/ab_libs.sh $1 3 "config3 --mode full --reps 300" default $V
bash scripts/ab_libs.sh $1 2 "config3 --mode fused --reps 300" default $V
bash scripts/ab_libs.sh $1 2 "config3 --mode full --policy ieee --reps 100" default $V
bash scripts/ab_libs.sh $1 2 "config2 --mode stats --reps 1000" default $V
bash scripts/ab_libs.sh $1 2 "config2 --mode full --reps 800" default $V
bash scripts/ab_libs.sh $1 2 "config5 --mode stats --reps 40" default $V
