set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
V=${1:-v6}
python bench.py > gpurun_out/bench_${V}_default.json 2> gpurun_out/bench_${V}_default.err
tail -1 gpurun_out/bench_${V}_default.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${V} -- python3 bench.py --no-cpu-baseline > gpurun_out/stats_${V}.log 2>&1
find gpurun_out/stats_${V} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r01_${V}_bench_kernel_stats.csv
head -5 gpurun_out/r01_${V}_bench_kernel_stats.csv
bash tools/pmc_traffic.sh ${V} 2>&1 | tail -4
for p in 1 2 3; do python tools/stage_profile.py $p s 2048 > gpurun_out/stage_r01_${V}_psy$p.txt 2>&1; done
python bench.py --no-cpu-baseline --psy 3 --streams 16384 2>/dev/null | tail -1 > gpurun_out/bench_r01_${V}_cfg3_psy3_16384.json
python bench.py --no-cpu-baseline --psy 2 2>/dev/null | tail -1 > gpurun_out/bench_r01_${V}_psy2.json
python bench.py --no-cpu-baseline --mode j 2>/dev/null | tail -1 > gpurun_out/bench_r01_${V}_psy1_joint.json
python bench.py --no-cpu-baseline --psy 0 --streams 16384 2>/dev/null | tail -1 > gpurun_out/bench_r01_${V}_psy0_16384.json
