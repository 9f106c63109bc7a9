# Round 5 (session 3): what the boundary of a SHORT timed region costs (the driver's command times 20 steps = 4 ms between two
# synchronisations): host-side wait settings of the ROCm runtime, the driver's command line four times each, same box.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_k; mkdir -p $OUT
run() {   # run <name> <env...>
  local NAME=$1; shift
  for i in 1 2 3 4; do
    env "$@" timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-other-configs 2>/dev/null | tail -1 > $OUT/${NAME}_$i.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/${NAME}_$i.json")); print("$NAME", $i, d["ms_per_step"], d["median_ms_per_step"], d["value"])
except Exception as e: print("$NAME", $i, "FAILED", e)
PY
  done
}
run base AIR_DUMMY=0
run nointr HSA_ENABLE_INTERRUPT=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=100000
run both HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=100000
run base2 AIR_DUMMY=0
# long runs under the candidate (does the steady state move?)
for V in base nointr; do
  E=AIR_DUMMY=0; [ $V = nointr ] && E=HSA_ENABLE_INTERRUPT=0
  env $E timeout 200 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --no-sweep --no-other-configs 2>/dev/null | tail -1 > $OUT/long_$V.json
  python -c "import json; d=json.load(open('$OUT/long_$V.json')); print('long $V', d['ms_per_step'], d['value'])"
done
