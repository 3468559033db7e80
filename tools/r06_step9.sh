#!/bin/bash
# round 6, GPU step 9: what the eight top subset lines of psy model 3's thresholds cost: the kernel without them against the kernel
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/pmc_quick.txt gpurun_out/ab_libs.txt
bash tools/pmc_quick.sh "--psy 3 --streams 16384 --frames-per-step 8" default build/lib_notail.so > /dev/null 2>&1; cat gpurun_out/pmc_quick.txt
bash tools/ab_libs.sh "--psy 3 --streams 16384 --frames-per-step 8" default build/lib_notail.so > /dev/null 2>&1; cat gpurun_out/ab_libs.txt
