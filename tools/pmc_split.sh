#!/bin/bash
# SQ counters per kernel of the split path (one rocprofv3 --pmc pass per counter group -- four groups --, kernel trace only).
# usage: tools/pmc_split.sh TAG [bench args...]     -> gpurun_out/sq_TAG.json
set -u
TAG=${1:-r02}; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
G2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
G3="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64"
# round 5: the instruction classes (what the bench line's issue-cost model weighs) and the lane occupancy of the vector instructions
G4="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sqs_${TAG}_$i
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/sqs_${TAG}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/sqs_${TAG}_$i.log 2>&1
done
python3 - "$TAG" "$@" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/sqs_{tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for name in ("tl_frame_kernel", "tl_psy2_kernel", "tl_main_kernel"):
            if name in k:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(f"gpurun_out/sqs_{tag}_*/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        for name in ("tl_frame_kernel", "tl_psy2_kernel", "tl_main_kernel"):
            if name in k:
                dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {"bench_args": sys.argv[2:], "note": "per launch; SQ cycle counters are in units of 4 clocks"}
for name, c in acc.items():
    d = {k: sum(v) / len(v) for k, v in c.items()}
    d["kernel_ms_under_pmc"] = sum(dur[name]) / max(1, len(dur[name]))
    out[name] = d
json.dump(out, open(f"gpurun_out/sq_{tag}.json", "w"), indent=1)
for name, d in out.items():
    if not isinstance(d, dict): continue
    wc = d.get("SQ_WAVE_CYCLES", 0)
    print(name, "ms", round(d["kernel_ms_under_pmc"], 4), "waves", d.get("SQ_WAVES"), "VALU", d.get("SQ_INSTS_VALU"), "SALU", d.get("SQ_INSTS_SALU"), "LDS", d.get("SQ_INSTS_LDS"), "VMEM", d.get("SQ_INSTS_VMEM"))
    if wc:
        print("   issuing", round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "waiting", round(d.get("SQ_WAIT_ANY", 0) / wc, 3), "valu/wave", round(d.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
              "wave_cycles/busy_cycles", round(wc / max(1.0, d.get("SQ_BUSY_CYCLES", 0)), 3), "gui", d.get("GRBM_GUI_ACTIVE"), "busy_cu", d.get("SQ_BUSY_CU_CYCLES"))
PY
