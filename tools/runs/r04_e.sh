#!/bin/bash
# round 4, call e: canvas + extreme tests on the fixed chain rule; old-vs-new canvas A/B; the bench line
O=gpurun_out/r04_e; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "canvas or st_write" > $O/canvas_tests.log 2>&1; echo "canvas tests rc=$?"; tail -8 $O/canvas_tests.log
timeout 1200 python -m pytest tests/test_extreme_scales.py -q -m gpu > $O/extreme.log 2>&1; echo "extreme rc=$?"; tail -12 $O/extreme.log
timeout 900 python tools/probes/canvas_ab.py > $O/canvas_ab.txt 2>&1; echo "ab rc=$?"; cat $O/canvas_ab.txt
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_e/bench_c2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline'))
for k in ('roofline_sweep_canvas_write_fwd','roofline_sweep_canvas_write_bwd','roofline_sweep_canvas_write_pair','roofline_sweep_st_read_fwd'):
    print(k, [(r.get('batch'), r.get('us_per_launch', r.get('us_fwd_plus_bwd')), r.get('frac')) for r in d.get(k,[])])
PY
