# Round 5, final tree, last check: bench.py's GPU tests + the driver's own command after the --step-bias probe flag went in
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u5; mkdir -p $OUT
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_u_bench_c2_b64_driver_command.json 2> $OUT/bench.log
python - <<PY
import json
d = json.loads(open("$OUT/r05_u_bench_c2_b64_driver_command.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["step_bias"], d["config"]["model_state_at_end"], d["other_configs"]["c4"]["ms_per_step"], d["other_configs"]["c5"]["ms_per_step"], d["cpu_baseline"]["value"])
PY
timeout 200 python -m pytest tests/test_bench_multirank.py -x -q -m gpu 2>&1 | tail -2
