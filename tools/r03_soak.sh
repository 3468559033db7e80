mkdir -p gpurun_out/r03soak
for seed in 401 402 403 404 405 406 407 408 409 410 411 412 413; do TL_SOAK_MODELS=2,4 timeout 600 python tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r03soak/soak24.txt 2>&1; done
for seed in 421 422 423 424 425 426 427 428; do timeout 600 python tools/soak_gpu.py 8192 16 $seed >> gpurun_out/r03soak/soak_all.txt 2>&1; done
cat gpurun_out/r03soak/soak24.txt gpurun_out/r03soak/soak_all.txt
