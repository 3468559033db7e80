#!/bin/bash
# round 6 step 13: the line loop's selects cut (no exchange of the predicted phase's pair, no zero cases in steps 1..7, min / max for the quotient's operands)
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz" 2>&1 | tail -3 ) > gpurun_out/r06_s13_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_base.so build/lib_split.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 4" build/lib_split.so default > /dev/null 2>&1
bash tools/pmc_quick.sh "--psy 2" default > /dev/null 2>&1
head -3 gpurun_out/r06_s13_gputests.log; cat gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
