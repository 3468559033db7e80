mkdir -p gpurun_out/r03c
python -m pytest tests -m gpu -x -q > gpurun_out/r03c/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c/pytest.txt
tail -5 gpurun_out/r03c/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()"
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r03c/bench_driver_style.json 2> gpurun_out/r03c/bench.err) 2>&1 | tail -3
python3 -c "
import json; d=json.loads(open('gpurun_out/r03c/bench_driver_style.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['output_check']['checked'], d['also']['configs2_psy3_16384'].get('traffic'))"
