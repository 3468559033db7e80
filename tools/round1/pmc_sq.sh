#!/bin/bash
# Where do the wave cycles go?  SQ counters for tl_encode_kernel (one --pmc pass per group).
set -u
TAG=${1:-r01}; PSY=${2:-1}
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
G2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --psy $PSY > gpurun_out/sq_${TAG}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/sq_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tl_encode_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc): print(f"{k:26s} {sum(acc[k])/len(acc[k]):16.0f}  (n={len(acc[k])})")
PY
