#!/bin/bash
# round 6, second soak of the FINAL kernels (other seeds, three times the volume of r06_soak.sh): random configurations and signals against the oracle, byte for byte
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
OUT=gpurun_out/soak_r06_final2.txt; : > $OUT
for seed in 6501 6502 6503 6504; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 16384 10 $seed >> $OUT 2>&1; done
for seed in 6511 6512 6513 6514; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
for seed in 6521 6522 6523; do timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> $OUT 2>&1; done
for seed in 6531 6532 6533; do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> $OUT 2>&1; done
for seed in 6541 6542; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
cat $OUT
