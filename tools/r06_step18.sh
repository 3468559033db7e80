#!/bin/bash
# round 6 step 18: glibc's sincos table in LDS with the rows' halves apart (16-byte gathers over sixteen bank quads instead of eight)
mkdir -p gpurun_out; rm -f gpurun_out/ab_libs.txt
( time python -m pytest tests -m gpu -x -q -k "psy or parity or golden or fuzz" 2>&1 | tail -3 ) > gpurun_out/r06_s18_gputests.log 2>&1
bash tools/ab_libs.sh "--psy 2" build/lib_prev.so default > /dev/null 2>&1
bash tools/ab_libs.sh "--config 4" build/lib_prev.so default > /dev/null 2>&1
head -3 gpurun_out/r06_s18_gputests.log; cat gpurun_out/ab_libs.txt
for L in build/lib_prev.so default; do
  if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$PWD/$L; fi
  rm -rf gpurun_out/ldsq; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $PWD/gpurun_out/ldsq -- python3 $PWD/bench.py --psy 2 --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2>&1
  python3 - "$L" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/ldsq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tl_psy2_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
d={k:sum(v)/len(v)/131072 for k,v in acc.items()}
print(sys.argv[1], "tl_psy2_kernel per frame:", {k:round(v,1) for k,v in d.items()})
PY
done
