#!/bin/bash
# round 6 step 16: the psy-2 kernel's next unit requested a unit ahead
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz or tick or node" 2>&1 | tail -3 ) > gpurun_out/r06_s16_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2 --streams 16384 --frames-per-step 1" build/lib_s15.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4 --frames-per-step 1" build/lib_s15.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_s15.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_s15.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2 --streams 1 --frames-per-step 8192" build/lib_s15.so default > /dev/null 2>&1
head -3 gpurun_out/r06_s16_gputests.log; cat gpurun_out/ab_libs.txt
