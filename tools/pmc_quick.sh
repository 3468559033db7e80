#!/bin/bash
# VALU / SALU / LDS instructions per frame, lane occupancy and VALU busy of tl_frame_kernel for one or more builds (one rocprofv3 --pmc pass each).
# usage: tools/pmc_quick.sh "bench args" lib.so [lib.so ...]   -> gpurun_out/pmc_quick.txt
set -u
ARGS=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for L in "$@"; do
  if [ "$L" = default ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/$L; fi
  T=$(basename $L .so); rm -rf gpurun_out/pq_$T
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pq_$T -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also $ARGS > gpurun_out/pq_$T.log 2>&1
  python3 - $T <<'PY' | tee -a gpurun_out/pmc_quick.txt
import csv, glob, sys, collections
t = sys.argv[1]; acc = collections.defaultdict(list); units = None
for f in glob.glob(f"gpurun_out/pq_{t}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tl_frame_kernel" in r["Kernel_Name"] or "tl_psy2_kernel" in r["Kernel_Name"] or "tl_main_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in acc})
for k in ks:
    d = {c: sum(v) / len(v) for (kk, c), v in acc.items() if kk == k}
    n = 131072.0
    gui = d.get("GRBM_GUI_ACTIVE", 0) / 8
    print(t, k, "VALU/frame %.0f SALU %.0f LDS %.0f lanes %.1f valu_busy_per_simd %.3f" % (d["SQ_INSTS_VALU"] / n, d["SQ_INSTS_SALU"] / n, d["SQ_INSTS_LDS"] / n,
          d["SQ_THREAD_CYCLES_VALU"] / max(1.0, d["SQ_ACTIVE_INST_VALU"]), d["SQ_ACTIVE_INST_VALU"] / (gui * 256) if gui else 0))
PY
done
