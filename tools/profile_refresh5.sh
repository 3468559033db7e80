#!/bin/bash
# Refresh of the headline files of a profiling round after a late kernel change (the light half of tools/profile_round5.sh):
# default bench line, kernel stats, HBM traffic and SQ counters of configs[1] and configs[2], stage stamps, class budget.  usage: tools/profile_refresh5.sh TAG
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p gpurun_out
V=${1:-r05}; R=$PWD; T="timeout 900"
$T python3 bench.py > gpurun_out/bench_${V}_default.json 2> gpurun_out/bench_${V}_default.err
tail -1 gpurun_out/bench_${V}_default.json | cut -c1-300
rm -rf gpurun_out/stats_${V}
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${V} -- python3 $R/bench.py --no-cpu-baseline --no-also > gpurun_out/stats_${V}.log 2>&1
find gpurun_out/stats_${V} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${V}_bench_kernel_stats.csv
head -4 gpurun_out/${V}_bench_kernel_stats.csv
for W in "hl" "psy3 --psy 3 --streams 16384 --frames-per-step 8" "mono --mode m"; do
  set -- $W; N=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${V}_${N}_$C
    $T rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${V}_${N}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > gpurun_out/pmc_${V}_${N}_$C.log 2>&1
  done
done
$T bash tools/pmc_split.sh ${V}_hl > gpurun_out/sq_${V}_hl.txt 2>&1
$T bash tools/pmc_split.sh ${V}_psy3 --psy 3 --streams 16384 --frames-per-step 8 > gpurun_out/sq_${V}_psy3.txt 2>&1
$T bash tools/pmc_split.sh ${V}_mono --mode m > gpurun_out/sq_${V}_mono.txt 2>&1
if ls build/lib_cb_EXP1.so > /dev/null 2>&1; then $T bash tools/class_budget.sh run > /dev/null 2>&1; cp gpurun_out/class_budget.txt gpurun_out/class_budget_${V}.txt; cat gpurun_out/class_budget_${V}.txt; fi
for p in 1 3; do $T python3 tools/stage_profile.py $p s 3072 > gpurun_out/stage_${V}_psy$p.txt 2>&1; done
$T python3 bench.py --no-cpu-baseline --no-also --psy 3 --streams 16384 --frames-per-step 8 2>/dev/null | tail -1 > gpurun_out/bench_${V}_cfg2_psy3_16384.json
$T python3 bench.py --no-cpu-baseline --no-also --psy 0 2>/dev/null | tail -1 > gpurun_out/bench_${V}_psy0.json
$T python3 bench.py --no-cpu-baseline --no-also --in-process 2 2>/dev/null | tail -1 > gpurun_out/bench_${V}_node2.json
for m in 1 3; do $T python3 bench.py --no-cpu-baseline --no-also --mode m --psy $m 2>/dev/null | tail -1 > gpurun_out/bench_${V}_mono_psy$m.json; done
