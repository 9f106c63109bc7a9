# Round 5, final tree: configs[4] / configs[3] / configs[1] timed near the INITIALISATION state (60 steps in all) next to the default runs, with the
# model state each ends in -- how much of the batch-1024 figure is the collapsed step count (profiles/r05_c4_seed_dependence.txt, item 5)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u3; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("model_state_at_end"))
PY
}
for C in c5 c4 c2; do
  for SW in "50 10" "200 300" "1000 100"; do
    set -- $SW
    timeout 200 python bench.py --config $C --no-cpu-baseline --no-sweep --no-other-configs --steps $1 --warmup $2 > $OUT/${C}_s$1_w$2.json 2>/dev/null
    show $OUT/${C}_s$1_w$2.json "$C steps=$1 warmup=$2"
  done
done | tee $OUT/states.txt
