#!/bin/bash
# Where do the vector instructions of tl_psy2_kernel (psy models 2 / 4) go, by STAGE and by CLASS, with each stage's lane occupancy and
# LDS bank conflicts?  Diagnostic builds that drop the last n stages of a full pass (TL_P2_LEVEL = 1..7, csrc/mp2_wave.h) or ONE
# operation of the line loop (TL_P2_SUB = 1..4), three counter passes each; the difference of successive builds is a stage's DYNAMIC
# count per stereo frame of the 4096 x 32 psy-2 workload (two channels x two passes; the seed passes of the cut chains are in the
# transform / polar rows).  Only the psy-2 translation unit is rebuilt, the other objects are the product's (build/obj).
#   tools/class_budget_psy2.sh build     (here: 11 libraries under build/)
#   tools/class_budget_psy2.sh run       (GPU box) -> gpurun_out/class_budget_psy2.txt   (copy to profiles/class_budget_rNN_psy2.txt)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
VARS="LEVEL:1 LEVEL:2 LEVEL:3 LEVEL:4 LEVEL:5 LEVEL:6 LEVEL:7 SUB:1 SUB:2 SUB:3 SUB:4"
if [ "${1:-run}" = build ]; then
  make -s -C odr-audioenc_amd/csrc > /dev/null 2>&1          # the product's objects
  mkdir -p build
  for v in $VARS; do
    k=${v%%:*}; n=${v##*:}; O=$R/build/obj_p2_${k}$n; mkdir -p $O
    ( cp $R/build/obj/*.o $O/ && rm -f $O/toolame_psy2.o
      make -s -C odr-audioenc_amd/csrc OBJ=$O EXTRA="-DTL_P2_${k}=$n -Wno-pass-failed ${EXTRA:-}" $O/toolame_psy2.o > /dev/null 2>&1
      cd odr-audioenc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread -Wl,--version-script=exports.map -o $R/build/lib_p2_${k}$n.so $O/*.o
      python3 $R/tools/check_isa.py $R/build/lib_p2_${k}$n.so --no-fail > $O/isa_guard.txt 2>&1 ) &
  done
  wait; ls build/ | grep -c "lib_p2_.*so$"
  exit 0
fi
export TMPDIR=/tmp; mkdir -p gpurun_out
W="${P2_WORKLOAD:---psy 2}"
G="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
G2="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
for v in BASE LEVEL1 LEVEL2 LEVEL3 LEVEL4 LEVEL5 LEVEL6 LEVEL7 SUB1 SUB2 SUB3 SUB4; do
  if [ $v = BASE ]; then unset TLB_LIB_PATH; else export TLB_LIB_PATH=$R/build/lib_p2_$v.so; [ -f $TLB_LIB_PATH ] || continue; fi
  i=0
  for g in "$G" "$G2" "$G3"; do
    i=$((i+1)); rm -rf gpurun_out/p2b${i}_$v
    timeout 120 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/p2b${i}_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also $W > gpurun_out/p2b${i}_$v.log 2>&1
  done
done
python3 - <<'PY' | tee gpurun_out/class_budget_psy2.txt
import csv, glob, collections
def load(v, i):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(f"gpurun_out/p2b{i}_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_psy2_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"].replace("SQ_INSTS_VALU_", "").replace("SQ_INSTS_", "").replace("SQ_", "")].append(float(row["Counter_Value"]))
    return {k: sum(x) / len(x) / 131072 for k, x in acc.items()}
def kernel_ms(v):
    t = []
    for f in glob.glob(f"gpurun_out/p2b1_{v}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tl_psy2_kernel" in row["Kernel_Name"]: t.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6)
    return sum(t) / len(t) if t else float("nan")
cols = ["VALU", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "CVT", "INT32", "INT64"]
def row(name, d, ms=None):
    f64 = d["ADD_F64"] + d["MUL_F64"] + d["FMA_F64"] + d["TRANS_F64"]
    other = d["VALU"] - f64 - d["CVT"] - d["INT32"] - d["INT64"]
    ratio = d["VALU"] / f64 if f64 > 0 else float("nan")
    print(f"{name:44s} {d['VALU']:7.0f} {f64:7.0f} {d['ADD_F64']:6.0f} {d['MUL_F64']:6.0f} {d['FMA_F64']:6.0f} {d['TRANS_F64']:6.0f} {d['CVT']:5.0f} {d['INT32']:6.0f} {d['INT64']:6.0f} {other:7.0f}   {ratio:5.2f}" + (f"   {ms:6.3f} ms" if ms is not None else ""))
print("# tl_psy2_kernel, 4096 streams x 32 frames, psy 2: dynamic VALU instructions per STEREO FRAME (2 channels x 2 passes) by stage and class")
print("# (SQ_INSTS_VALU_* counters under rocprofv3, diagnostic builds, tools/class_budget_psy2.sh).  f64 = ADD + MUL + FMA + TRANS; other = VALU - f64 - CVT - INT32 -")
print("# INT64 (moves, selects, compares, bit operations, lane reads).  A stage's row = the build that still has it minus the build that dropped it;")
print("# 'ms' = the kernel's duration under the profiler for the build that still HAS the stage (whole kernel) resp. the time the stage's removal saves.")
print(f"{'stage':44s} {'VALU':>7s} {'f64':>7s} {'add':>6s} {'mul':>6s} {'fma':>6s} {'trans':>6s} {'cvt':>5s} {'int32':>6s} {'int64':>6s} {'other':>7s}   VALU/f64")
base = load("BASE", 1); bms = kernel_ms("BASE")
row("whole kernel", base, bms)
sub = lambda a, b: {k: a.get(k, 0) - b.get(k, 0) for k in a}
names = ["32 subbands (min / sum of 17 lines, log)", "per-line thresholds", "spreading 64x64 + SNR + permissible noise", "partition sums (energy, weighted c)",
         "unpredictability: 2 sincos + c[] per line", "polar form: energy, sqrt, atan2 per line", "window + transform (FHT)"]
prev, pms = base, bms
for n in range(1, 8):
    cur = load(f"LEVEL{n}", 1)
    if not cur: print("LEVEL", n, "no data"); continue
    ms = kernel_ms(f"LEVEL{n}")
    row(names[n - 1], sub(prev, cur), pms - ms); prev, pms = cur, ms
row("(skeleton: units, state load / store)", prev, pms)
print("# single operations of the line loop (the build without it against the whole kernel):")
for n, nm in ((1, "sincos of the predicted phase"), (2, "both sincos"), (3, "atan2"), (4, "2 sqrt + 1 division")):
    cur = load(f"SUB{n}", 1)
    if cur: row("  " + nm, sub(base, cur), bms - kernel_ms(f"SUB{n}"))
print()
print("# lane occupancy of the vector instructions (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)), scalar / LDS / VMEM instructions, LDS conflicts per stage")
print(f"{'stage':44s} {'VALU':>7s} {'lanes':>6s} {'SALU':>6s} {'LDS':>6s} {'VMEM':>6s} {'LDSact':>7s} {'bankcf':>7s} {'cf %':>6s}")
def both(v):
    a = load(v, 2); b = load(v, 3)
    if not a: return None
    a.update({k: x for k, x in b.items() if k not in a}); return a
def orow(name, d):
    lanes = d["THREAD_CYCLES_VALU"] / d["ACTIVE_INST_VALU"] if d.get("ACTIVE_INST_VALU", 0) > 0 else float("nan")
    act = d.get("LDS_IDX_ACTIVE", 0); bc = d.get("LDS_BANK_CONFLICT", 0)
    print(f"{name:44s} {d['VALU']:7.0f} {lanes:6.1f} {d['SALU']:6.0f} {d['LDS']:6.0f} {d['VMEM']:6.0f} {act:7.0f} {bc:7.0f} {100 * bc / act if act else 0:6.1f}")
ob = both("BASE")
if ob:
    orow("whole kernel", ob)
    prev = ob
    for n in range(1, 8):
        cur = both(f"LEVEL{n}")
        if not cur: continue
        orow(names[n - 1], sub(prev, cur)); prev = cur
    orow("(skeleton: units, state load / store)", prev)
PY
