#!/bin/bash
# Dynamic VALU instruction mix by class per kernel (SQ_INSTS_VALU_* counters, one rocprofv3 --pmc pass per group, kernel trace only).
# usage: tools/pmc_classes.sh TAG [bench args...]     -> gpurun_out/classes_TAG.json   (TLB_LIB_PATH selects an experimental build)
set -u
TAG=${1:-r04}; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
G1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
G2="SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_SALU SQ_INSTS_BRANCH"
G3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/cls_${TAG}_$i
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/cls_${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/cls_${TAG}_$i.log 2>&1
done
python3 - "$TAG" "$@" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = ("tl_frame_kernel", "tl_psy2_kernel", "tl_main_kernel", "tl_psy2f_kernel")
for f in glob.glob(f"gpurun_out/cls_{tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for name in names:
            if name in k:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"bench_args": sys.argv[2:], "note": "per launch, average over the launches of the run"}
for name, c in acc.items():
    out[name] = {k: sum(v) / len(v) for k, v in c.items()}
json.dump(out, open(f"gpurun_out/classes_{tag}.json", "w"), indent=1)
for name, d in out.items():
    if not isinstance(d, dict): continue
    w = d.get("SQ_WAVES", 1)
    print(name, {k.replace("SQ_INSTS_", "").replace("SQ_", ""): round(v) for k, v in sorted(d.items())})
PY
