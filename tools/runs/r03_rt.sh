#!/bin/bash
# the runtime setting applied by the package itself (no shell export): bench lines of every named configuration, GPU suite, smoke
O=gpurun_out/r03_q; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r03_gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r03_q bench > $O/profile_bench.log 2>&1
timeout 300 python bench.py --steps-per-replay 4 --no-cpu-baseline --no-sweep > $O/r03_q_bench_c2_b64_4steps_per_replay.json 2>> $O/bench.log
AIR_RUNTIME_ENV=0 timeout 300 python bench.py --no-cpu-baseline --no-sweep > $O/r03_q_bench_c2_b64_runtime_defaults.json 2>> $O/bench.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_q/r03_q_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("traffic"), d["config"].get("hip_runtime_env"))
    except Exception as e: print(f, "ERR", e)
PY
