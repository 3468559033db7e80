# round 5 soak of the committed kernels (run on the GPU box): random configurations and signals against the oracle, byte for byte
mkdir -p gpurun_out/r05soak
OUT=gpurun_out/r05soak/soak_${1:-a}.txt
: > $OUT
S0=${2:-0}; for seed in $((911+S0)) $((912+S0)) $((913+S0)) $((914+S0)); do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> $OUT 2>&1; done
for seed in $((921+S0)) $((922+S0)); do timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
for seed in $((931+S0)) $((932+S0)); do TL_SOAK_EDGE=1 TL_SOAK_MODELS=1,3,2,4 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> $OUT 2>&1; done
for seed in $((941+S0)) $((942+S0)); do TL_SOAK_MODELS=0,1,3 timeout 900 python3 tools/soak_gpu.py 16384 6 $seed >> $OUT 2>&1; done
grep -c "0 mismatching" $OUT; grep -v "0 mismatching" $OUT | head
