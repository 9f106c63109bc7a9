# Round 5, final tree: the driver's own command once more after `median_block_ms_per_step` joined other_configs (the record of the final bench.py)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u6; mkdir -p $OUT
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_u_bench_c2_b64_driver_command.json 2> $OUT/bench.log
python - <<PY
import json
d = json.loads(open("$OUT/r05_u_bench_c2_b64_driver_command.json").read().strip().splitlines()[-1])
oc = d["other_configs"]
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], "| c4", oc["c4"]["ms_per_step"], oc["c4"]["median_block_ms_per_step"], [(p["ms_per_step"], p["median_block_ms_per_step"]) for p in oc["c4"]["per_seed"]], "| c5", oc["c5"]["ms_per_step"], oc["c5"]["median_block_ms_per_step"])
PY
