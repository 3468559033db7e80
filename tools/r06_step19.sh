#!/bin/bash
# round 6 step 19: VERDICT r5 item 3 (a), the literals re-made inside loops: the main translation unit WITH machine LICM (build/lib_licm.so: the frame kernels then
# spill 32-84 vector registers to scratch, the encoder-only kernel does not) against the product
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
bash tools/ab_libs.sh "" default build/lib_licm.so > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 0" default build/lib_licm.so > /dev/null 2>&1
bash tools/pmc_quick.sh "" default build/lib_licm.so > /dev/null 2>&1
bash tools/pmc_quick.sh "--psy 0" default build/lib_licm.so > /dev/null 2>&1
cat gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
