#!/bin/bash
# round 6, GPU step 7: the transform's passes with all operands read before the first store (models 1 / 3) -- whole GPU suite, A/B
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > gpurun_out/r06_s7_gputests.log 2>&1; tail -6 gpurun_out/r06_s7_gputests.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "" build/lib_fhtold.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 3 --streams 16384 --frames-per-step 8" build/lib_fhtold.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--mode m" build/lib_fhtold.so default > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
timeout 300 python3 tools/stage_profile.py 1 s 3072 2>&1 | grep "FHT\|spectrum"
