# Round 5, final tree: the two ends of the state-dependent cost -- every step present (--step-bias 20) / none (--step-bias -20) at the three
# single-GPU configurations, 400 timed steps behind 100, with the model state of the last step
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u4; mkdir -p $OUT
for C in c2 c4 c5; do for SB in 20 -20; do
  timeout 200 python bench.py --config $C --step-bias $SB --no-cpu-baseline --no-sweep --no-other-configs --steps 400 --warmup 100 > $OUT/${C}_sb$SB.json 2>/dev/null
  python - $OUT/${C}_sb$SB.json "$C step_bias=$SB" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("model_state_at_end"), d["config"]["params_finite_after_run"])
PY
done; done | tee $OUT/ends.txt
