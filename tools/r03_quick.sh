# quick check on the GPU box: psy 2/4 parity tests + their bench lines (+ anything passed as arguments)
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "known_bad or degenerate or soak_slice or golden_all or configuration_sweep" 2>&1 | tail -3
for p in 2 4; do python bench.py --no-cpu-baseline --no-also --psy $p --steps 30 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('psy', $p, d['value'], d['ms_per_step'], d['roofline'].get('kernels_ms'), d['output_check'].get('checked'))"; done
"$@"
