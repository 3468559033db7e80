#!/bin/bash
# round 6, GPU step 1: fault-isolation tests, device parity of the psy-2 kernel's spreading-by-bands, A/B against the round-5 library
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_node_fault_gpu.py tests/test_node_gpu.py -x -q -m gpu > gpurun_out/r06_s1_fault.log 2>&1; tail -3 gpurun_out/r06_s1_fault.log
timeout 1500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or psy2 or psy4 or configs4 or sweep or soak or mono" > gpurun_out/r06_s1_parity.log 2>&1; tail -3 gpurun_out/r06_s1_parity.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "--psy 2" build/lib_r05.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_r05.so default > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
bash tools/pmc_quick.sh "--psy 2" default > /dev/null 2>&1; cat gpurun_out/pmc_quick.txt | tail -3
