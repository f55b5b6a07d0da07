#!/bin/bash
# GPU box, round 2 step 1: gpu tests, FAST attribution table, A/B against the round-1 library.
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 600 python scripts/fast_attribution.py 2024 300 > $OUT/fast_attr.log 2>&1; tail -25 $OUT/fast_attr.log
timeout -k 10 600 bash scripts/ab_variants.sh 2>&1 | tail -8
