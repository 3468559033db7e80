#!/bin/bash
# One profiling round on the GPU box (round 6).  bash tools/profile_round6.sh TAG  -> gpurun_out/, then (here) tools/install_profiles6.py TAG
# rocprofv3 runs the program itself after `--` (python3 bench.py ...); counters are collected in their own passes with --kernel-trace
# only (never with the sys/hip/hsa trace domains).  Needs build/lib_cb_*.so (tools/class_budget.sh build) for step 5.
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p gpurun_out
V=${1:-r06}
R=$PWD
T="timeout 900"
# 1. the default bench line (what the driver runs), with `also` and the CPU baseline
$T python3 bench.py > gpurun_out/bench_${V}_default.json 2> gpurun_out/bench_${V}_default.err
tail -1 gpurun_out/bench_${V}_default.json | cut -c1-300
# 2. kernel trace + stats of the same command (no secondary measurements, no CPU leg: same kernels, same shapes)
rm -rf gpurun_out/stats_${V}
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${V} -- python3 $R/bench.py --no-cpu-baseline --no-also > gpurun_out/stats_${V}.log 2>&1
find gpurun_out/stats_${V} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${V}_bench_kernel_stats.csv
head -6 gpurun_out/${V}_bench_kernel_stats.csv
# 3. HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes: configs[1], configs[2] (= one GPU of configs[3]), psy 2, configs[4] with psy 4 and psy 2, mono pairs
for W in "hl" "psy3 --psy 3 --streams 16384 --frames-per-step 8" "psy2 --psy 2" "cfg4psy4 --config 4" "cfg4psy2 --config 4 --psy 2" "mono --mode m" \
         "tick2 --psy 2 --streams 16384 --frames-per-step 1" "tick3 --psy 3 --streams 16384 --frames-per-step 1" "cfg4tick --config 4 --frames-per-step 1"; do
  set -- $W; N=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${V}_${N}_$C
    $T rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${V}_${N}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/pmc_${V}_${N}_$C.log 2>&1
  done
done
# 4. SQ counters per kernel (three passes each)
$T bash tools/pmc_split.sh ${V}_hl > gpurun_out/sq_${V}_hl.txt 2>&1
$T bash tools/pmc_split.sh ${V}_psy3 --psy 3 --streams 16384 --frames-per-step 8 > gpurun_out/sq_${V}_psy3.txt 2>&1
$T bash tools/pmc_split.sh ${V}_psy2 --psy 2 > gpurun_out/sq_${V}_psy2.txt 2>&1
$T bash tools/pmc_split.sh ${V}_cfg4psy4 --config 4 > gpurun_out/sq_${V}_cfg4psy4.txt 2>&1
$T bash tools/pmc_split.sh ${V}_mono --mode m > gpurun_out/sq_${V}_mono.txt 2>&1
$T bash tools/pmc_split.sh ${V}_tick2 --psy 2 --streams 16384 --frames-per-step 1 > gpurun_out/sq_${V}_tick2.txt 2>&1
cat gpurun_out/sq_${V}_hl.txt
# 5. dynamic VALU mix by stage and class (diagnostic builds) and by kernel; round 6: the same for the psy-2 kernel (tools/class_budget_psy2.sh build)
if ls build/lib_p2_LEVEL1.so > /dev/null 2>&1; then $T bash tools/class_budget_psy2.sh run > /dev/null 2>&1; cp gpurun_out/class_budget_psy2.txt gpurun_out/class_budget_${V}_psy2.txt; cat gpurun_out/class_budget_${V}_psy2.txt; fi
if ls build/lib_cb_EXP1.so > /dev/null 2>&1; then $T bash tools/class_budget.sh run > /dev/null 2>&1; cp gpurun_out/class_budget.txt gpurun_out/class_budget_${V}.txt; cat gpurun_out/class_budget_${V}.txt; fi
$T bash tools/pmc_classes.sh ${V}_psy2 --psy 2 > gpurun_out/classes_${V}_psy2.txt 2>&1
# 6. where a frame's time goes (s_memtime stamps)
for p in 1 3 2; do $T python3 tools/stage_profile.py $p s 3072 > gpurun_out/stage_${V}_psy$p.txt 2>&1; done
# 7. other workloads, one line each
$T python3 bench.py --no-cpu-baseline --no-also --psy 2 2>/dev/null | tail -1 > gpurun_out/bench_${V}_psy2.json
$T python3 bench.py --no-cpu-baseline --no-also --psy 4 2>/dev/null | tail -1 > gpurun_out/bench_${V}_psy4.json
$T python3 bench.py --no-cpu-baseline --no-also --psy 0 2>/dev/null | tail -1 > gpurun_out/bench_${V}_psy0.json
$T python3 bench.py --no-cpu-baseline --no-also --psy 3 --streams 16384 --frames-per-step 8 2>/dev/null | tail -1 > gpurun_out/bench_${V}_cfg2_psy3_16384.json
$T python3 bench.py --no-cpu-baseline --no-also --frames-per-step 8 2>/dev/null | tail -1 > gpurun_out/bench_${V}_cfg1_F8.json
$T python3 bench.py --no-cpu-baseline --config 4 2>/dev/null | tail -1 > gpurun_out/bench_${V}_cfg4.json
$T python3 bench.py --no-cpu-baseline --no-also --psy 2 --streams 16384 --frames-per-step 1 2>/dev/null | tail -1 > gpurun_out/bench_${V}_tick2.json
$T python3 bench.py --no-cpu-baseline --no-also --config 4 --frames-per-step 1 2>/dev/null | tail -1 > gpurun_out/bench_${V}_cfg4tick.json
$T python3 bench.py --no-cpu-baseline --no-also --in-process 2 2>/dev/null | tail -1 > gpurun_out/bench_${V}_node2.json
$T python3 bench.py --no-cpu-baseline --no-also --backend nccl --force-group 2>/dev/null | tail -1 > gpurun_out/bench_${V}_rccl1.json
for m in 1 3 0 2; do $T python3 bench.py --no-cpu-baseline --no-also --mode m --psy $m 2>/dev/null | tail -1 > gpurun_out/bench_${V}_mono_psy$m.json; done
$T python3 tools/legacy_latency.py 2000 1 > gpurun_out/legacy_latency_${V}.txt 2>&1
$T python3 tools/legacy_latency.py 2000 2 >> gpurun_out/legacy_latency_${V}.txt 2>&1
cat gpurun_out/legacy_latency_${V}.txt
# 8. round 6: the whole GPU suite on the same kernels (fault isolation included), its log kept
( time timeout 2400 python3 -m pytest tests -q -m gpu ) > gpurun_out/gputests_${V}_final.log 2>&1; tail -5 gpurun_out/gputests_${V}_final.log
ls gpurun_out | grep ${V} | head -80
# 9. the driver's smoke on the same box
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_${V}.log 2>&1; tail -2 gpurun_out/smoke_${V}.log
