# Round 5, final tree: whole GPU suite, smoke, then the default line and the driver's own command (tag r05_u: same kernels as r05_t -- the
# digest-matched in-step JSONs of r05_t apply -- with configs[3] aggregated over three engine seeds and `model_state_at_end` in every line).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u; mkdir -p $OUT
timeout 800 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 300 python bench.py > $OUT/r05_u_bench_c2_b64_unprofiled.json 2> $OUT/bench.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_u_bench_c2_b64_driver_command.json 2>> $OUT/bench.log
timeout 200 python bench.py --config c4 --no-cpu-baseline --steps 1000 --warmup 100 > $OUT/r05_u_bench_c4_b64.json 2>> $OUT/bench.log
python - <<PY
import json
for n in ("c2_b64_unprofiled", "c2_b64_driver_command", "c4_b64"):
    d = json.loads(open("$OUT/r05_u_bench_%s.json" % n).read().strip().splitlines()[-1])
    oc = d.get("other_configs") or {}
    print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"].get("model_state_at_end"), "| c4", (oc.get("c4") or {}).get("ms_per_step"), [p["ms_per_step"] for p in (oc.get("c4") or {}).get("per_seed", [])], "| c5", (oc.get("c5") or {}).get("ms_per_step"))
PY
