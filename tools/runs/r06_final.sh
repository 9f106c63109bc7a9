# Round 6, shipped build: profiler passes (tag $1), then -- with the digest-matched PMC / in-step / MFMA-utilisation JSONs in profiles/ -- the
# unprofiled lines of every named configuration (the default line and the driver's own command first), then the whole GPU suite and smoke().
ROOT=$(pwd); TAG=${1:-r06_x}; mkdir -p gpurun_out/$TAG
bash tools/profile_round.sh $TAG pmc > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/$TAG/${TAG}_*_instep_pmc.json gpurun_out/$TAG/${TAG}_*_instep_durations.json gpurun_out/$TAG/${TAG}_*_mfma_util.json profiles/
bash tools/profile_round.sh $TAG bench > gpurun_out/${TAG}_bench.log 2>&1
python - <<PY
import json
for n in ("c2_b64_unprofiled", "c2_b64_driver_command", "c4_b64", "c5_b1024_bf16", "c2_b1024_f32"):
    try:
        d = json.loads(open("gpurun_out/$TAG/${TAG}_bench_%s.json" % n).read().strip().splitlines()[-1])
        oc = d.get("other_configs") or {}
        print(n, d["ms_per_step"], d["value"], "launches", d["config"]["kernel_launches_per_step"], "fixed", (d.get("fixed_batch") or {}).get("ms_per_step"),
              "roofline", d["roofline"].get("frac"), d["roofline"].get("traffic"), "gemm", d["roofline_gemm"].get("frac"), (d["roofline_gemm"].get("mfma_busy_utilisation") or {}).get("util_of_busy"),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "| c4", (oc.get("c4") or {}).get("ms_per_step"), (oc.get("c4") or {}).get("per_seed_spread"), "| c5", (oc.get("c5") or {}).get("ms_per_step"))
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 1600 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/$TAG/gpu_tests.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/$TAG/smoke.log
