#!/bin/bash
# HBM traffic of one GPU's share of BASELINE configs[4] (tools/bench_configs4.py: 16384 streams, 32 kHz mono 64 k / 48 kHz stereo 192 k
# interleaved, psy 4 and psy 2): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, kernel trace only.   usage: tools/pmc_configs4.sh TAG
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; mkdir -p gpurun_out
V=${1:-r03c}; R=$PWD
for P in 4 2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${V}_cfg4psy${P}_$C
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${V}_cfg4psy${P}_$C -- python3 $R/tools/bench_configs4.py $P > gpurun_out/pmc_${V}_cfg4psy${P}_$C.log 2>&1
  done
done
ls gpurun_out | grep cfg4psy
