#!/bin/bash
# GPU box: A/B builds re-checked at SUSTAINED clocks (hundreds of back-to-back calls; the 8-call A/Bs of this round ran in the
# clock's ramp): statistics-only config 3, config 5, config 4, config-2 summary
cd /root/repo
V=build/variants
bash scripts/ab_libs.sh r4_ab_sustained.log 2 "config3 --mode stats --reps 300" default $V/libort_wt16.so $V/libort_wt4.so $V/libort_pil.so
bash scripts/ab_libs.sh r4_ab_sustained.log 2 "config5 --mode stats --reps 40" default $V/libort_wt16.so $V/libort_wt4.so
bash scripts/ab_libs.sh r4_ab_sustained.log 2 "config4 --reps 80" default $V/libort_sw64.so
bash scripts/ab_libs.sh r4_ab_sustained.log 2 "config2 --mode summary --reps 1500" default $V/libort_sw64.so
