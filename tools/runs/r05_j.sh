# Round 5 (session 3): XCD-aware tile placement of the latency-regime grouped GEMMs (AIR_GEMM_MCOLOC = 0 / 1 / 2), same-box A/B at
# configs[1] and configs[3]; GEMM + engine parity under the widest setting first.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_j; mkdir -p $OUT
AIR_GEMM_MCOLOC=2 timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_engine.py -x -q -m gpu -k "gemm or linear or forward_and_gradients or graph_replay or updates_match" > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -2
for i in 1 2; do
  for V in 0 1 2; do
    AIR_GEMM_MCOLOC=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/c2_m${V}_$i.json
    AIR_GEMM_MCOLOC=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_m${V}_$i.json
    python - <<PY
import json
for c in ("c2", "c4"):
    d=json.load(open("$OUT/%s_m${V}_$i.json" % c)); print(c, "mcoloc=$V run $i", d["ms_per_step"], d["value"])
PY
  done
done
