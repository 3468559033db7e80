#!/bin/bash
# Kernel A/B on one box: tools/ab_libs.sh "bench args" lib1.so lib2.so ...  (3 interleaved rounds; TLB_LIB_PATH selects the build)
# prints frames/s and the HIP-event kernel time of every run -> gpurun_out/ab_libs.txt
set -u
ARGS=$1; shift
mkdir -p gpurun_out
for r in 1 2 3; do
  for L in "$@"; do
    if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$PWD/$L; fi
    python3 bench.py --no-also --no-cpu-baseline --steps 100 --warmup 10 $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', 'round $r', d['value'], 'frames/s', d['roofline']['kernel_ms'], 'ms', d['output_check'].get('checked'))"
  done
done | tee -a gpurun_out/ab_libs.txt
