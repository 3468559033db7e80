#!/bin/bash
# Where do the VALU instructions of the psy phase of tl_frame_kernel go?  Diagnostic builds (TL_EXP_LEVEL = n removes the last n stages of psy model 1)
# under one SQ counter pass each.  Build the variants HERE first (tools/instr_budget.sh build), run on the GPU box (… run).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
if [ "${1:-run}" = build ]; then
  mkdir -p build
  for n in 1 2 3 4 5 6 7 8; do
    make -s -C odr-audioenc_amd/csrc OUT=$R/build/lib_exp$n.so OBJ=$R/build/obj_exp$n ISA_GUARD=--no-fail EXTRA="-DTL_EXP_LEVEL=$n -Wno-pass-failed" > /dev/null 2>&1 &
    if [ $((n % 4)) = 0 ]; then wait; fi
  done
  wait; ls -la build/
  exit 0
fi
export TMPDIR=/tmp; mkdir -p gpurun_out
for n in 0 1 2 3 4 5 6 7 8; do
  if [ $n = 0 ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/build/lib_exp$n.so; fi
  rm -rf gpurun_out/ib_$n
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/ib_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/ib_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
names = ["(all stages)", "- thresholds", "- decimation", "- chains/weights/centres", "- noise compaction", "- tone walk/levels", "- candidates", "- power+spike", "- spectrum (window+FHT)"]
prev = None
for n in range(9):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(f"gpurun_out/ib_{n}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_frame_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/ib_{n}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_frame_kernel" in row["Kernel_Name"]: dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    if not acc: print(n, "no data"); continue
    v = {k: sum(x) / len(x) / 131072 for k, x in acc.items()}
    ms = sum(dur) / len(dur)
    d = "" if prev is None else f"   stage: VALU {prev[0] - v['SQ_INSTS_VALU']:7.0f}  SALU {prev[1] - v['SQ_INSTS_SALU']:6.0f}  LDS {prev[2] - v['SQ_INSTS_LDS']:6.0f}  ms {prev[3] - ms:6.3f}"
    print(f"level {n} {names[n]:28s} per frame: VALU {v['SQ_INSTS_VALU']:7.0f} SALU {v['SQ_INSTS_SALU']:6.0f} LDS {v['SQ_INSTS_LDS']:6.0f}  kernel {ms:6.3f} ms{d}")
    prev = (v['SQ_INSTS_VALU'], v['SQ_INSTS_SALU'], v['SQ_INSTS_LDS'], ms)
PY
