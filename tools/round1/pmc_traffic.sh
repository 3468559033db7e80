#!/bin/bash
# HBM traffic of tl_encode_kernel from the PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE need
# separate --pmc passes (TCC slots).  Run on the GPU box from the repo root: bash tools/pmc_traffic.sh <tag>
set -u
TAG=${1:-r01}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -1 gpurun_out/pmc_${TAG}_$C.log | cut -c1-200
done
python3 - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("gpurun_out/pmc_${TAG}_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_encode_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
                vals.append(float(row["Counter_Value"]))
    print(c, "launches", len(vals), "mean per launch", sum(vals) / max(1, len(vals)))
PY
