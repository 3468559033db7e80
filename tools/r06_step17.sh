#!/bin/bash
# round 6 step 17: the transform's butterflies dealt to the lanes by LDS bank (host table only; k=4: 28 lanes of a half shared 8 banks, k=6: halves straddled two blocks)
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt gpurun_out/pmc_quick.txt
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r06_s17_gputests.log 2>&1
bash tools/ab_libs.sh "" build/lib_prof2.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 3 --streams 16384 --frames-per-step 8" build/lib_prof2.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_prof2.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--mode m" build/lib_prof2.so default > /dev/null 2>&1
head -3 gpurun_out/r06_s17_gputests.log; cat gpurun_out/ab_libs.txt
for L in build/lib_prof2.so default; do
  if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$PWD/$L; fi
  rm -rf gpurun_out/ldsq; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $PWD/gpurun_out/ldsq -- python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2>&1
  python3 - "$L" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/ldsq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tl_frame_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
d={k:sum(v)/len(v)/131072 for k,v in acc.items()}
print(sys.argv[1], "per frame:", {k:round(v,1) for k,v in d.items()})
PY
done
