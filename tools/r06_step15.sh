#!/bin/bash
# round 6 step 15: the psy-2 run's prediction state fetched by its first pass (behind the transform's loads) instead of at the head of the unit
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz or tick or node" 2>&1 | tail -3 ) > gpurun_out/r06_s15_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2 --streams 16384 --frames-per-step 1" build/lib_prof.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4 --frames-per-step 1" build/lib_prof.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_prof.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_prof.so default > /dev/null 2>&1
head -3 gpurun_out/r06_s15_gputests.log; cat gpurun_out/ab_libs.txt
