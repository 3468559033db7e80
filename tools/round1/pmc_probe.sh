#!/bin/bash
# usage: pmc_probe.sh "<bench args>" COUNTER [COUNTER...]   -- mean per launch of tl_encode_kernel
export TMPDIR=/tmp; R=$PWD; ARGS=$1; shift
D=$R/gpurun_out/probe_$$
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$D/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "tl_encode" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$ARGS", {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
rm -rf $D
