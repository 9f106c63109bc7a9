#!/bin/bash
# the committed bench lines regenerated with the round's last bench.py (same build, same runtime setting)
O=gpurun_out/r03_q; mkdir -p $O
bash tools/profile_round.sh r03_q bench > $O/profile_bench.log 2>&1
timeout 300 python bench.py --steps-per-replay 4 --no-cpu-baseline --no-sweep > $O/r03_q_bench_c2_b64_4steps_per_replay.json 2>> $O/bench.log
AIR_RUNTIME_ENV=0 timeout 300 python bench.py --no-cpu-baseline --no-sweep > $O/r03_q_bench_c2_b64_runtime_defaults.json 2>> $O/bench.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_q/r03_q_bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("traffic"), d["config"].get("replicas_in_sync_after_run"))
PY
