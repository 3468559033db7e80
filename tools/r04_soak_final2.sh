mkdir -p gpurun_out/r04soak
: > gpurun_out/r04soak/soak_final2.txt
for seed in 811 812 813 814 815 816 817 818; do timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak_final2.txt 2>&1; done
for seed in 821 822 823 824; do TL_SOAK_MODELS=2,4 timeout 900 python3 tools/soak_gpu.py 8192 24 $seed >> gpurun_out/r04soak/soak_final2.txt 2>&1; done
for seed in 831 832 833 834; do TL_SOAK_MODELS=1,3 timeout 900 python3 tools/soak_gpu.py 16384 9 $seed >> gpurun_out/r04soak/soak_final2.txt 2>&1; done
for seed in 841 842 843 844; do TL_SOAK_EDGE=1 TL_SOAK_MODELS=2,4,1 timeout 900 python3 tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r04soak/soak_final2.txt 2>&1; done
for seed in 851 852 853 854; do TL_SOAK_MODELS=0,1,3 timeout 900 python3 tools/soak_gpu.py 16384 6 $seed >> gpurun_out/r04soak/soak_final2.txt 2>&1; done
grep -c "0 mismatching" gpurun_out/r04soak/soak_final2.txt; grep -v "0 mismatching" gpurun_out/r04soak/soak_final2.txt | head
