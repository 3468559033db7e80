mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03b/pytest.txt
tail -15 gpurun_out/r03b/pytest.txt
python tools/legacy_latency.py > gpurun_out/r03b/legacy.txt 2>&1; cat gpurun_out/r03b/legacy.txt
timeout 900 python bench.py > gpurun_out/r03b/bench.json 2> gpurun_out/r03b/bench.err; tail -c 3000 gpurun_out/r03b/bench.json; tail -5 gpurun_out/r03b/bench.err
