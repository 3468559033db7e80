#!/bin/bash
# round 6, GPU step 3: the whole GPU suite on the current kernels, stage stamps (psy 2 and the new running-order rows of psy 1 / 3), A/B
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > gpurun_out/r06_s3_gputests.log 2>&1; tail -6 gpurun_out/r06_s3_gputests.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "--psy 2" build/lib_r05.so default build/lib_sched.so > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_r05.so default build/lib_sched.so > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
for L in build/lib_r05.so default; do
  if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$PWD/$L; fi
  echo "== $L"; timeout 300 python3 tools/stage_profile.py 2 s 3072 2>&1 | grep -v "warning\|Warning" | head -30
done | tee gpurun_out/r06_s3_stage_psy2.txt
unset TLB_LIB_PATH
for p in 1 3; do timeout 300 python3 tools/stage_profile.py $p s 3072 > gpurun_out/stage_r06_psy$p.txt 2>&1; done; cat gpurun_out/stage_r06_psy1.txt
rm -f gpurun_out/pmc_quick.txt; bash tools/pmc_quick.sh "--psy 2" default > /dev/null 2>&1; cat gpurun_out/pmc_quick.txt | tail -2
