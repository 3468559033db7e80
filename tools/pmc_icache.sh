#!/bin/bash
# Instruction-cache counters per kernel (one rocprofv3 --pmc pass, kernel trace only).
# usage: tools/pmc_icache.sh TAG [bench args...]     -> gpurun_out/icache_TAG.txt
set -u
TAG=${1:-r02}; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_PREFETCH|INST_LEVEL)[A-Z_0-9]*)\b" | sort -u > gpurun_out/icache_avail_$TAG.txt
cat gpurun_out/icache_avail_$TAG.txt
G="SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
rm -rf $R/gpurun_out/ic_$TAG
rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/ic_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/ic_$TAG.log 2>&1
python3 - "$TAG" <<'PY' | tee gpurun_out/icache_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/ic_{tag}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for name in ("tl_frame_kernel", "tl_psy2_kernel", "tl_main_kernel"):
            if name in k:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, c in acc.items():
    print(name, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
tail -3 gpurun_out/ic_$TAG.log
