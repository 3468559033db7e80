#!/bin/bash
# round 6 step 11: psy 2's partition sums unrolled by two (u2only) and the padded subband walk on top (default) against the step-10 kernel (base)
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz" 2>&1 | tail -3 ) > gpurun_out/r06_s11_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_base.so build/lib_u2only.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 4" build/lib_base.so default > /dev/null 2>&1
bash tools/pmc_quick.sh "--psy 2" build/lib_base.so default > /dev/null 2>&1
python tools/stage_profile.py --psy 2 > gpurun_out/r06_s11_stage_psy2.txt 2>&1
tail -3 gpurun_out/r06_s11_gputests.log; cat gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt; head -24 gpurun_out/r06_s11_stage_psy2.txt
