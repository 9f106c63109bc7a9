# Round 5, shipped build: profiler passes (tag r05_s), then -- with the digest-matched PMC / in-step JSONs in profiles/ -- the unprofiled
# lines of every named configuration, then the whole GPU suite.
ROOT=$(pwd); TAG=${1:-r05_s}
EVERY=2 bash tools/profile_round.sh $TAG pmc > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/$TAG/${TAG}_*_instep_pmc.json gpurun_out/$TAG/${TAG}_*_instep_durations.json profiles/
bash tools/profile_round.sh $TAG bench > gpurun_out/${TAG}_bench.log 2>&1
python - <<PY
import json
for n in ("c2_b64_unprofiled", "c2_b64_driver_command", "c4_b64", "c5_b1024_bf16", "c2_b1024_f32"):
    try:
        d = json.loads(open("gpurun_out/$TAG/${TAG}_bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("traffic"), (d.get("cpu_baseline") or {}).get("all_host_cores_replicas"))
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 1600 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/$TAG/gpu_tests.log | tail -2
