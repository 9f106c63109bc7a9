# Round 5 (session 3): the whole GPU suite on the tree as restored, then a same-box A/B of the LSTM weight gradients' placement
# (AIR_LSTM_DW_EARLY=0/1, alternating, unprofiled headline lines), then the default line and the driver's command.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_i; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for i in 1 2 3; do
  for V in 0 1; do
    AIR_LSTM_DW_EARLY=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_early${V}_$i.json
    python - <<PY
import json; d=json.load(open("$OUT/bench_early${V}_$i.json")); print("early=$V run $i", d["ms_per_step"], d["value"], d.get("kernel_launches"))
PY
  done
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2>> $OUT/bench_default.log
python - <<PY
import json
for n in ("default", "driver"):
    d = json.loads(open("$OUT/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], {k: (v.get("value"), v.get("ms_per_step")) for k, v in (d.get("other_configs") or {}).items()})
PY
