#!/bin/bash
# Where do the VALU instructions of tl_frame_kernel<1> go, by STAGE and by CLASS?  Diagnostic builds that drop the last n stages of the
# psy-1 model phase (TL_EXP_LEVEL = 1..7, 9 = no model) or of the encoder phase (TL_ENC_LEVEL = 1..5), one SQ_INSTS_VALU_* counter pass each; the
# difference of successive builds is a stage's DYNAMIC instruction count per class, per stereo frame of BASELINE configs[1].
#   tools/class_budget.sh build     (here: 13 libraries under build/)
#   tools/class_budget.sh run       (GPU box) -> gpurun_out/class_budget.txt   (copy to profiles/class_budget_rNN.txt)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
if [ "${1:-run}" = build ]; then
  mkdir -p build; i=0
  for v in EXP:1 EXP:2 EXP:3 EXP:4 EXP:5 EXP:6 EXP:7 EXP:9 ENC:1 ENC:2 ENC:3 ENC:4 ENC:5; do
    k=${v%%:*}; n=${v##*:}; i=$((i+1))
    make -s -C odr-audioenc_amd/csrc OUT=$R/build/lib_cb_${k}$n.so OBJ=$R/build/obj_cb_${k}$n ISA_GUARD=--no-fail EXTRA="-DTL_${k}_LEVEL=$n -Wno-pass-failed ${EXTRA:-}" > /dev/null 2>&1 &
    if [ $((i % 5)) = 0 ]; then wait; fi
  done
  wait; ls build/ | grep lib_cb | wc -l
  exit 0
fi
export TMPDIR=/tmp; mkdir -p gpurun_out
G2="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM"
G="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
for v in BASE EXP1 EXP2 EXP3 EXP4 EXP5 EXP6 EXP7 EXP9 ENC1 ENC2 ENC3 ENC4 ENC5; do
  if [ $v = BASE ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/build/lib_cb_$v.so; fi
  rm -rf gpurun_out/cb_$v
  timeout 120 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/cb_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/cb_$v.log 2>&1
  # second pass: lane occupancy of the vector instructions (thread-cycles over instruction-cycles x 64) and the LDS / scalar counts
  rm -rf gpurun_out/cbo_$v
  timeout 120 rocprofv3 --pmc $G2 --kernel-trace --output-format csv -d $R/gpurun_out/cbo_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/cbo_$v.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/class_budget.txt
import csv, glob, collections
def load(v, pre="cb"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{pre}_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_frame_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"].replace("SQ_INSTS_VALU_", "").replace("SQ_INSTS_", "")].append(float(row["Counter_Value"]))
    return {k: sum(x) / len(x) / 131072 for k, x in acc.items()}
def occ(v):
    d = {k[3:] if k.startswith("SQ_") else k: x for k, x in load(v, "cbo").items()}
    return d if d and "THREAD_CYCLES_VALU" in d else None
cols = ["VALU", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "CVT", "INT32", "INT64"]
OCC = {}
def row(name, d, o=None):
    if o: OCC[name] = o
    f64 = d["ADD_F64"] + d["MUL_F64"] + d["FMA_F64"] + d["TRANS_F64"]
    other = d["VALU"] - f64 - d["CVT"] - d["INT32"] - d["INT64"]
    ratio = d["VALU"] / f64 if f64 > 0 else float("nan")
    print(f"{name:34s} {d['VALU']:7.0f} {f64:7.0f} {d['ADD_F64']:6.0f} {d['MUL_F64']:6.0f} {d['FMA_F64']:6.0f} {d['CVT']:6.0f} {d['INT32']:6.0f} {d['INT64']:6.0f} {other:7.0f}   {ratio:5.2f}")
print("# tl_frame_kernel<1>, 4096 streams x 32 frames (configs[1]), dynamic VALU instructions per stereo frame by stage and class (SQ_INSTS_VALU_* counters,")
print("# diagnostic builds, tools/class_budget.sh).  f64 = ADD + MUL + FMA + TRANS;  other = VALU - f64 - CVT - INT32 - INT64 (moves, selects, compares,")
print("# bit operations, lane reads, DPP).  A stage's row = the build that still has it minus the build that dropped it.")
print(f"{'stage':34s} {'VALU':>7s} {'f64':>7s} {'add':>6s} {'mul':>6s} {'fma':>6s} {'cvt':>6s} {'int32':>6s} {'int64':>6s} {'other':>7s}   VALU/f64")
base = load("BASE")
row("whole frame", base)
sub = lambda a, b: {k: a[k] - b[k] for k in cols}
names = ["thresholds", "decimation", "dB-sum chains + weights + centres", "noise compaction", "tone walk + levels", "tone candidates", "power spectrum + spike levels", "spectrum (window + FHT)"]
prev = base
for n in range(1, 8):
    cur = load(f"EXP{n}")
    if not cur: print("EXP", n, "no data"); continue
    row("psy: " + names[n - 1], sub(prev, cur)); prev = cur
enc = load("EXP9")          # no model at all: the encoder phase (the build without the spectrum alone is not meaningful: the stages after it then chew on garbage)
if enc:
    row("psy: " + names[7], sub(prev, enc))
    row("(encoder phase: the model removed)", enc)
names = ["CRC-16 + ScF-CRC + X-PAD", "quantiser + sample packing", "header / bit_alloc / scf fields", "bit allocation", "scalefactors + SMR + pattern"]
prev = base
for n in range(1, 6):
    cur = load(f"ENC{n}")
    if not cur: print("ENC", n, "no data"); continue
    row("enc: " + names[n - 1], sub(prev, cur)); prev = cur
row("(psy phase + filterbank + staging)", prev)
# ---- lane occupancy by stage: thread-cycles of the vector instructions over (instruction-cycles x 64) ----
print()
print("# lane occupancy of the vector instructions by stage (second counter pass per build: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64);")
print("# both in units of 4 cycles per wave-instruction), with the stage's scalar and LDS instruction counts per frame")
print(f"{'stage':34s} {'VALU':>7s} {'lanes':>6s} {'SALU':>6s} {'LDS':>6s} {'VMEM':>6s}")
ocols = ["THREAD_CYCLES_VALU", "ACTIVE_INST_VALU", "VALU", "SALU", "LDS", "VMEM"]
def orow(name, d):
    lanes = d["THREAD_CYCLES_VALU"] / d["ACTIVE_INST_VALU"] if d["ACTIVE_INST_VALU"] > 0 else float("nan")
    print(f"{name:34s} {d['VALU']:7.0f} {lanes:6.1f} {d['SALU']:6.0f} {d['LDS']:6.0f} {d['VMEM']:6.0f}")
osub = lambda a, b: {k: a[k] - b[k] for k in ocols}
ob = occ("BASE")
if ob:
    orow("whole frame", ob)
    names = ["thresholds", "decimation", "dB-sum chains + weights + centres", "noise compaction", "tone walk + levels", "tone candidates", "power spectrum + spike levels", "spectrum (window + FHT)"]
    prev = ob
    for n in range(1, 8):
        cur = occ(f"EXP{n}")
        if not cur: continue
        orow("psy: " + names[n - 1], osub(prev, cur)); prev = cur
    e9 = occ("EXP9")
    if e9:
        orow("psy: " + names[7], osub(prev, e9)); orow("(encoder phase: the model removed)", e9)
    names = ["CRC-16 + ScF-CRC + X-PAD", "quantiser + sample packing", "header / bit_alloc / scf fields", "bit allocation", "scalefactors + SMR + pattern"]
    prev = ob
    for n in range(1, 6):
        cur = occ(f"ENC{n}")
        if not cur: continue
        orow("enc: " + names[n - 1], osub(prev, cur)); prev = cur
    orow("(psy phase + filterbank + staging)", prev)
PY
