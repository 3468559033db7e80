#!/bin/bash
# round 6 step 12: psy 2's partition sums as pure addition chains, two lanes per partition (default) against step 11's kernel (pad) and step 10's (base)
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz" 2>&1 | tail -3 ) > gpurun_out/r06_s12_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_base.so build/lib_pad.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 4" build/lib_pad.so default > /dev/null 2>&1
bash tools/pmc_quick.sh "--psy 2" default > /dev/null 2>&1
python tools/stage_profile.py 2 > gpurun_out/r06_s12_stage_psy2.txt 2>&1
tail -3 gpurun_out/r06_s12_gputests.log; cat gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt; head -24 gpurun_out/r06_s12_stage_psy2.txt
