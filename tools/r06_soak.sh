#!/bin/bash
# round 6 soak of the committed kernels (run on the GPU box): random configurations and signals against the oracle, byte for byte
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
OUT=gpurun_out/soak_r06_final.txt; : > $OUT
for seed in 6401 6402; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 16384 10 $seed >> $OUT 2>&1; done
for seed in 6411 6412; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
for seed in 6421; do timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> $OUT 2>&1; done
for seed in 6431; do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> $OUT 2>&1; done
cat $OUT
