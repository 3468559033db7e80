python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden_all or configuration_sweep or known_bad or fuzz" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-also --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('psy1', d['value'], d['ms_per_step'], d['output_check'].get('checked'))"
python bench.py --no-cpu-baseline --no-also --steps 100 --psy 3 --streams 16384 --frames-per-step 8 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('psy3', d['value'], d['ms_per_step'], d['output_check'].get('checked'))"
