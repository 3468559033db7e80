mkdir -p gpurun_out/r04soak
: > gpurun_out/r04soak/soak_final.txt
for seed in 711 712 713 714; do timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak_final.txt 2>&1; done
for seed in 721 722; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 24 $seed >> gpurun_out/r04soak/soak_final.txt 2>&1; done
for seed in 731 732; do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> gpurun_out/r04soak/soak_final.txt 2>&1; done
for seed in 741 742; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=2,4,1 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak_final.txt 2>&1; done
grep -c "0 mismatching" gpurun_out/r04soak/soak_final.txt; grep -v "0 mismatching" gpurun_out/r04soak/soak_final.txt | head
