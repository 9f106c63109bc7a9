#!/bin/bash
# final build of the round: full GPU suite, probes, PMC evidence, then the bench lines with the digest-matched PMC JSONs in place
O=gpurun_out/r03_p; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r03_gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 300 python tools/probes/canvas_scaling.py 2>&1 | grep -v amdgpu.ids > $O/canvas_scaling.txt
for S in "0.45 0.65" "0.9 1.0"; do echo "== where scales $S"; timeout 120 tools/kbench/bin/st_trace 64 3 50 20 1 4 $S 2>&1 | grep -A13 -E "^canvas_unroll_bwd|^canvas_fused|^canvas_unroll_fwd_banded"; done > $O/st_trace.txt
bash tools/profile_round.sh r03_p pmc > $O/profile_pmc.log 2>&1
cp $O/r03_p_*_instep_pmc.json profiles/
bash tools/profile_round.sh r03_p bench > $O/profile_bench.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_p/r03_p_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("traffic"))
    except Exception as e: print(f, "ERR", e)
PY
