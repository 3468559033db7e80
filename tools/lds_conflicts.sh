#!/bin/bash
# LDS bank conflicts by STAGE of tl_frame_kernel<1>: the diagnostic builds of tools/class_budget.sh (build/lib_cb_*.so), one counter pass each
# (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_INSTS_LDS per frame); a stage's row = the build that still has it minus the build that dropped it.
#   tools/lds_conflicts.sh   (GPU box) -> gpurun_out/lds_conflicts.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
G="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES"
for v in BASE EXP1 EXP2 EXP3 EXP4 EXP5 EXP6 EXP7 EXP9 ENC1 ENC2 ENC3 ENC4 ENC5; do
  if [ $v = BASE ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/build/lib_cb_$v.so; fi
  rm -rf gpurun_out/lc_$v
  timeout 120 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/lc_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/lc_$v.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/lds_conflicts.txt
import csv, glob, collections
def load(v):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/lc_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_frame_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(x) / len(x) / 131072 for k, x in acc.items()}
names = ["thresholds", "decimation", "dB-sum chains + weights + centres", "noise compaction", "tone walk + levels", "tone candidates", "power spectrum + spike levels", "spectrum (window + FHT)"]
enc = ["CRC-16 + ScF-CRC + X-PAD", "quantiser + sample packing", "header / bit_alloc / scf fields", "bit allocation", "scalefactors + SMR + pattern"]
cols = ["SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"]
def row(n, d):
    act = d.get("SQ_LDS_IDX_ACTIVE", 0); bc = d.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{n:36s} " + " ".join(f"{d.get(c, 0):10.0f}" for c in cols) + f"   {100 * bc / act if act else 0:5.1f} %")
print("# per stereo frame: LDS instructions, cycles the LDS index path is active, of them bank-conflict cycles (+ address conflicts, unaligned stalls), share")
print(f"{'stage':36s} " + " ".join(f"{c.replace('SQ_', '')[:10]:>10s}" for c in cols) + "   conflicts")
base = load("BASE"); row("whole frame", base)
sub = lambda a, b: {k: a.get(k, 0) - b.get(k, 0) for k in cols}
prev = base
for n in range(1, 8):
    cur = load(f"EXP{n}")
    if cur: row("psy: " + names[n - 1], sub(prev, cur)); prev = cur
e9 = load("EXP9")
if e9:
    row("psy: " + names[7], sub(prev, e9)); row("(encoder phase: the model removed)", e9)
    prev = base                                  # the ENC builds keep the model and drop encoder stages from the end
    for n in range(1, 6):
        cur = load(f"ENC{n}")
        if cur: row("enc: " + enc[n - 1], sub(prev, cur)); prev = cur
    row("(psy phase + filterbank + staging)", prev)
    row("filterbank + staging + unit glue", sub(prev, sub(base, e9)))
PY
