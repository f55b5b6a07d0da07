#!/bin/bash
# GPU box: summary kernels counting the per-surface status only on demand (default) against always (-DORT_STATUS_ON_DEMAND=0), sustained clocks
cd /root/repo
V=build/variants/libort_nsd.so
bash scripts/ab_libs.sh $1 3 "config4 --reps 80" default $V
bash scripts/ab_libs.sh $1 2 "config2 --mode summary --reps 1500" default $V
bash scripts/ab_libs.sh $1 2 "config5 --mode hits --reps 30" default $V
bash scripts/ab_libs.sh $1 2 "config2 --mode history --reps 1500" default $V
