# Round 5, last session: the default line with configs[3] aggregated over three engine seeds (bench.run_other_config_seeds) + the model
# state each seed ends in; then the driver's own command, for the record of the final tree.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_z4; mkdir -p $OUT
{
for S in 1 1000004 7 12345; do for V in 1 0; do
  AIR_PROBE_SEED=$S AIR_GEMM_SHORTK=$V timeout 120 python tools/probes/placement_probe.py c4 fresh 2>/dev/null | tail -2 | cut -d'|' -f1
done; done
} | tee $OUT/seeds_state.txt
timeout 300 python bench.py > $OUT/r05_u_bench_c2_b64_unprofiled.json 2> $OUT/bench.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_u_bench_c2_b64_driver_command.json 2>> $OUT/bench.log
python - <<PY
import json
for n in ("unprofiled", "driver_command"):
    d = json.loads(open("$OUT/r05_u_bench_c2_b64_%s.json" % n).read().strip().splitlines()[-1])
    oc = d["other_configs"]
    print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], "| c4", oc["c4"].get("ms_per_step"), oc["c4"].get("value"), oc["c4"].get("per_seed"), "| c5", oc["c5"].get("ms_per_step"), oc["c5"].get("model_state_at_end"))
PY
timeout 600 python -m pytest tests/test_bench_multirank.py tests/test_engine.py -x -q -m gpu -k "bench or tf_checkpoint or launches_per" 2>&1 | tail -3
