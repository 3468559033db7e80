#!/bin/bash
# round 6, GPU step 5: partition sums pipelined, next pass's PCM touched early -- parity, soak slice, A/B
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or psy2 or psy4 or configs4 or sweep or soak or mono or known_bad or degenerate or api_shapes or ragged or one_stream" > gpurun_out/r06_s5_parity.log 2>&1; tail -3 gpurun_out/r06_s5_parity.log
OUT=gpurun_out/soak_r06_b.txt; : > $OUT
for seed in 6201; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
tail -2 $OUT
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "--psy 2" build/lib_r05.so default build/lib_nopipe.so build/lib_notouch.so > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_r05.so default build/lib_nopipe.so build/lib_notouch.so > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2 --streams 16384 --frames-per-step 1" build/lib_r05.so default > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
