#!/bin/bash
# round 6, GPU step 2: device parity of the psy-2 kernel's new arithmetic forms, A/B of its variants, stage stamps
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "golden or psy2 or psy4 or configs4 or sweep or soak or mono or known_bad or degenerate" > gpurun_out/r06_s2_parity.log 2>&1; tail -3 gpurun_out/r06_s2_parity.log
rm -f gpurun_out/ab_libs.txt
bash tools/ab_libs.sh "--psy 2" build/lib_r05.so default build/lib_noband.so build/lib_nons.so > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_r05.so default build/lib_noband.so > /dev/null 2>&1
cat gpurun_out/ab_libs.txt
for L in build/lib_r05.so default build/lib_noband.so; do
  if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$PWD/$L; fi
  echo "== $L"; timeout 300 python3 tools/stage_profile.py 2 s 3072 2>&1 | grep -v "^  \|^psy" | head -20
done | tee gpurun_out/r06_s2_stage.txt
unset TLB_LIB_PATH
rm -f gpurun_out/pmc_quick.txt; bash tools/pmc_quick.sh "--psy 2" default build/lib_noband.so > /dev/null 2>&1; cat gpurun_out/pmc_quick.txt | tail -4
