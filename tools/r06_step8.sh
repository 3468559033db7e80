#!/bin/bash
# round 6, GPU step 8: psy-2 kernel with the state's oldest quarter parked in LDS during the transform + reads-first butterflies -- A/B; node / bench tests
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_node_gpu.py tests/test_node_fault_gpu.py -x -q -m gpu -s > gpurun_out/r06_s8_node.log 2>&1; tail -3 gpurun_out/r06_s8_node.log; grep -m2 "devices_observed\|^shard 0" gpurun_out/r06_s8_node.log
timeout 900 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "two_ranks" > gpurun_out/r06_s8_ranks.log 2>&1; tail -2 gpurun_out/r06_s8_ranks.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "--psy 2" build/lib_r05.so default build/lib_park2.so > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" default build/lib_park2.so > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
