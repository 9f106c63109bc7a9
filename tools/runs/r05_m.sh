# Round 5 (session 3): the BPTT entry folded into its first link (air_lstm_step_bwd_entry, AIR_LSTM_BWD_ENTRY=0/1): kernel + engine parity,
# then a same-box A/B at configs[1] and configs[3]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine.py tests/test_abi_exports.py tests/test_api.py -x -q -m gpu -k "lstm or forward_and_gradients or graph_replay or updates_match or large_batch or riders or folded or abi or switches or argument" > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/tests.log | head -20
for i in 1 2 3; do
  for V in 0 1; do
    AIR_LSTM_BWD_ENTRY=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/c2_e${V}_$i.json
    AIR_LSTM_BWD_ENTRY=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_e${V}_$i.json
    python - <<PY
import json
for c in ("c2", "c4"):
    d=json.load(open("$OUT/%s_e${V}_$i.json" % c)); print(c, "entry=$V run $i", d["ms_per_step"], d["value"], d["kernel_launches_per_step"])
PY
  done
done
