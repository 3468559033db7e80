# GPU soak of round 3 (see profiles/soak_r03.txt): usage  bash tools/r03_soak.sh [tag]
T=${1:-final}
mkdir -p gpurun_out/r03soak
for seed in 511 512 513 514 515 516; do TL_SOAK_MODELS=2,4 timeout 600 python tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r03soak/soak24_$T.txt 2>&1; done
for seed in 521 522 523 524; do timeout 900 python tools/soak_gpu.py 8192 24 $seed >> gpurun_out/r03soak/soak_all_$T.txt 2>&1; done
cat gpurun_out/r03soak/soak24_$T.txt gpurun_out/r03soak/soak_all_$T.txt
