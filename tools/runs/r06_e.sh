OUT=gpurun_out/r06_e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_extreme_scales.py -q -m gpu -k "canvas or st_write or extreme or grid_stride" > $OUT/canvas_tests.log 2>&1; tail -3 $OUT/canvas_tests.log; grep -E "^(FAILED|ERROR)" $OUT/canvas_tests.log | head -40
timeout 900 python tools/probes/canvas_gs_ab.py > $OUT/canvas_gs_ab.txt 2>&1; cat $OUT/canvas_gs_ab.txt
for gs in 0 1; do
  AIR_CANVAS_BWD_GS=$gs timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 > $OUT/bench_c2_gs$gs.json 2>> $OUT/bench.log
done
python - <<PY
import json
for gs in (0, 1):
    d = json.loads(open("$OUT/bench_c2_gs%d.json" % gs).read().strip().splitlines()[-1])
    oc = d.get("other_configs") or {}
    print("gs", gs, "c2", d["ms_per_step"], "| c4", (oc.get("c4") or {}).get("ms_per_step"), [p["ms_per_step"] for p in (oc.get("c4") or {}).get("per_seed", [])], "| c5", (oc.get("c5") or {}).get("ms_per_step"))
PY
