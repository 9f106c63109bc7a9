# configs[3] (T = 5): the fused canvas launch (and the folds that come with it) against the plain plan, same box
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_q; mkdir -p $OUT
for i in 1 2 3; do
  for V in auto 1; do
    AIR_FUSE_CANVAS=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_f${V}_$i.json
    python - <<PY
import json
d=json.load(open("$OUT/c4_f${V}_$i.json")); print("c4 fuse_canvas=$V run $i", d["ms_per_step"], d["value"], d["config"].get("kernel_launches_per_step"))
PY
  done
done
