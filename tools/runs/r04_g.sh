#!/bin/bash
# round 4, call g: the whole GPU suite on the round's kernels + the bench line
O=gpurun_out/r04_g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -12 $O/gpu_tests.log
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_g/bench_c2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
PY
