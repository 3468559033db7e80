# psy 0/1/3 (what DAB services use) on the committed kernels: bash tools/r03_soak13.sh
mkdir -p gpurun_out/r03soak
for seed in $(seq 801 830); do TL_SOAK_MODELS=0,1,1,3,3 timeout 600 python tools/soak_gpu.py 8192 24 $seed >> gpurun_out/r03soak/soak13.txt 2>&1; done
grep -c " 0 mismatching" gpurun_out/r03soak/soak13.txt; grep -v " 0 mismatching" gpurun_out/r03soak/soak13.txt | head
