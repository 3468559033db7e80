#!/bin/bash
# Shader clock, power and temperature of the GPU while the default bench workload runs (rocm-smi sampled every 0.5 s) -> gpurun_out/clock_probe.txt
# Question it answers: is the encode kernel's clock (rocprofv3 says ~2.17 GHz of the 2.4 GHz peak) held down by the power limit?
mkdir -p gpurun_out
OUT=gpurun_out/clock_probe.txt
rocm-smi --showclocks --showpower --showtemp --showperflevel > $OUT 2>&1
echo "---- sampling during: bench.py --no-also --no-cpu-baseline --steps 1500 $*" >> $OUT
python3 bench.py --no-also --no-cpu-baseline --steps 1500 --warmup 10 "$@" > gpurun_out/clock_probe_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.5
done
wait $BP
tail -c 300 gpurun_out/clock_probe_bench.json >> $OUT
