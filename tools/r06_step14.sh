#!/bin/bash
# round 6 step 14: the transform's butterfly addresses from a host-built table, exchange partners as (byte offset ^ constant) + base in one v_xad_u32
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r06_s14_gputests.log 2>&1
bash tools/ab_libs.sh "" build/lib_sel.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 3 --streams 16384 --frames-per-step 8" build/lib_sel.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_sel.so default > /dev/null 2>&1
bash tools/pmc_quick.sh "" build/lib_sel.so default > /dev/null 2>&1
bash tools/pmc_quick.sh "--psy 2" default > /dev/null 2>&1
head -3 gpurun_out/r06_s14_gputests.log; cat gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
