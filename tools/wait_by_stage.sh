#!/bin/bash
# How much of each STAGE's wave time of tl_frame_kernel<1> is spent in s_waitcnt?  The diagnostic builds of tools/class_budget.sh (build/lib_cb_*.so), one
# counter pass each (SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_WAIT_INST_LDS, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU); a stage's row = the build that still has it minus the
# build that dropped it.  A stage whose waves wait half of their cycles while all three waves of a SIMD run the same code is where requesting loads a phase
# ahead pays (the psy-2 line loop, round 6); a stage at the pure-issue figure is not.
#   tools/wait_by_stage.sh   (GPU box) -> gpurun_out/wait_by_stage.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
G="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
for v in BASE EXP1 EXP2 EXP3 EXP4 EXP5 EXP6 EXP7 EXP9 ENC1 ENC2 ENC3 ENC4 ENC5; do
  if [ $v = BASE ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/build/lib_cb_$v.so; [ -f $TLB_LIB_PATH ] || continue; fi
  rm -rf gpurun_out/ws_$v
  timeout 120 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/ws_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/ws_$v.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/wait_by_stage.txt
import csv, glob, collections
def load(v):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/ws_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_frame_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"].replace("SQ_", "")].append(float(row["Counter_Value"]))
    return {k: sum(x) / len(x) / 131072 for k, x in acc.items()}
names = ["thresholds", "decimation", "dB-sum chains + weights + centres", "noise compaction", "tone walk + levels", "tone candidates", "power spectrum + spike levels", "spectrum (window + FHT)"]
enc = ["CRC-16 + ScF-CRC + X-PAD", "quantiser + sample packing", "header / bit_alloc / scf fields", "bit allocation", "scalefactors + SMR + pattern"]
def row(n, d):
    wc = d.get("WAVE_CYCLES", 0)
    print(f"{n:38s} {d.get('INSTS_VALU', 0):7.0f} {wc:9.0f} {100 * d.get('WAIT_INST_ANY', 0) / wc if wc else 0:7.1f} {100 * d.get('WAIT_INST_LDS', 0) / wc if wc else 0:7.1f} {100 * d.get('ACTIVE_INST_VALU', 0) / wc if wc else 0:7.1f} {wc / d['INSTS_VALU'] * 4 if d.get('INSTS_VALU') else 0:7.1f}")
print("# tl_frame_kernel<1>, configs[1]: per stereo frame and stage -- vector instructions, wave cycles (x4 clocks), share of them waiting in s_waitcnt (any / LDS),")
print("# share issuing a vector instruction, and wave clocks per vector instruction (three waves share a SIMD: 12-13 = pure issue)")
print(f"{'stage':38s} {'VALU':>7s} {'wavecyc':>9s} {'wait %':>7s} {'lds %':>7s} {'valu %':>7s} {'clk/VALU':>8s}")
base = load("BASE"); row("whole frame", base)
sub = lambda a, b: {k: a.get(k, 0) - b.get(k, 0) for k in a}
prev = base
for n in range(1, 8):
    cur = load(f"EXP{n}")
    if cur: row("psy: " + names[n - 1], sub(prev, cur)); prev = cur
e9 = load("EXP9")
if e9:
    row("psy: " + names[7], sub(prev, e9)); row("(encoder phase: the model removed)", e9)
    prev = base
    for n in range(1, 6):
        cur = load(f"ENC{n}")
        if cur: row("enc: " + enc[n - 1], sub(prev, cur)); prev = cur
    row("filterbank + staging + unit glue", sub(prev, sub(base, e9)))
PY
