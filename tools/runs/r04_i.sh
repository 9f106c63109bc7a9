#!/bin/bash
# round 4, call i: bench.py's N > 1 protocol A/B + watchdog on a shared GPU; the N = 1 line with the in-step roofline
O=gpurun_out/r04_i; mkdir -p $O
timeout 1500 python -m pytest tests/test_bench_multirank.py tests/test_distributed_cpu.py -q -m gpu -x > $O/multirank.log 2>&1; echo "multirank rc=$?"; tail -25 $O/multirank.log
timeout 900 python bench.py --steps 200 --warmup 20 --no-sweep --cpu-seconds 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -3 $O/bench_c2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_i/bench_c2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['us_per_launch'], d['roofline']['frac'], d['roofline']['kernel'][:40], d['cpu_baseline'])
PY
