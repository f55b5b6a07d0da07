cd /root/repo
V=build/variants
bash scripts/ab_libs.sh r4_ab_tune.log 2 "config5 --mode stats --reps 40" default $V/libort_wt8k.so $V/libort_wt32k.so $V/libort_f32w6.so $V/libort_f32w4.so
bash scripts/ab_libs.sh r4_ab_tune.log 2 "config3 --mode stats --reps 300" default $V/libort_wt8k.so $V/libort_wt32k.so
