mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest.txt
tail -5 gpurun_out/r03a/pytest.txt
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; tail -c 1500 gpurun_out/r03a/bench.json
for seed in 301 302 303 304 305 306 307; do TL_SOAK_MODELS=2,4 timeout 600 python tools/soak_gpu.py 8192 12 $seed >> gpurun_out/r03a/soak24.txt 2>&1; done
cat gpurun_out/r03a/soak24.txt
