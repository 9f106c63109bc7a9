# Round 5, last session: why does configs[3] read 0.324 ms as `other_configs.c4` of the default line and 0.304 ms as `--config c4`
# on the same box and binary (r05_t)?  Streaming short-K weight gradients on / off x (default line with sweeps | without sweeps | standalone).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_y; mkdir -p $OUT
timeout 300 python -m pytest tests/test_engine.py -x -q -m gpu -k "tf_checkpoint or checkpoint_resume" > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/tests.log | head -20
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d.get("other_configs") or {}
print(sys.argv[2], "headline", d["ms_per_step"], d["value"], "| c4", (oc.get("c4") or {}).get("ms_per_step"), "| c5", (oc.get("c5") or {}).get("ms_per_step"))
PY
}
for i in 1 2; do
  for V in 1 0; do
    AIR_GEMM_SHORTK=$V timeout 200 python bench.py --no-cpu-baseline > $OUT/line_sweeps_s${V}_$i.json 2>/dev/null; show $OUT/line_sweeps_s${V}_$i.json "shortk=$V with sweeps   run $i"
    AIR_GEMM_SHORTK=$V timeout 200 python bench.py --no-cpu-baseline --no-sweep > $OUT/line_nosweep_s${V}_$i.json 2>/dev/null; show $OUT/line_nosweep_s${V}_$i.json "shortk=$V without sweeps run $i"
    AIR_GEMM_SHORTK=$V timeout 200 python bench.py --config c4 --no-cpu-baseline --no-sweep --no-other-configs --steps 400 --warmup 100 > $OUT/c4_alone_s${V}_$i.json 2>/dev/null; show $OUT/c4_alone_s${V}_$i.json "shortk=$V --config c4 alone  run $i"
  done
done
