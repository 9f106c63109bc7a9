OUT=gpurun_out/r06_g; mkdir -p $OUT
T=tools/kbench/bin/st_trace
( echo "== c4 b2048 narrow"; $T 2048 5 100 28 1 4 0.45 0.65; echo "== c2 b4096 narrow"; $T 4096 3 50 20 1 4 0.45 0.65 ) > $OUT/st_trace_im.txt 2>&1
grep -B1 -A11 "trace canvas_unroll_bwd  " $OUT/st_trace_im.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver.log; tail -3 $OUT/bench_driver.log
timeout 600 python bench.py --no-cpu-baseline --no-sweep > $OUT/bench_default_nosweep.json 2> $OUT/bench_default.log; tail -3 $OUT/bench_default.log
python - <<PY
import json
for n in ("bench_driver_cmd", "bench_default_nosweep"):
    d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    oc = d.get("other_configs") or {}
    print(n, d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["config"].get("model_state_at_end"), "| fixed", d.get("fixed_batch"))
    for k, v in oc.items():
        print("  ", k, v.get("ms_per_step"), v.get("model_state_at_end"), [ (p["ms_per_step"], p["model_state_at_end"]["steps_present_per_image"]) for p in v.get("per_seed", [])], "fixed", v.get("fixed_batch"), "probe", v.get("all_steps_present_probe"), v.get("error"))
        for kk in ("roofline_sweep_st_read_fwd", "roofline_sweep_canvas_write_fwd", "roofline_sweep_canvas_write_bwd", "roofline_sweep_canvas_write_pair"):
            if kk in v: print("     ", kk, [(x["batch"], x["frac"]) for x in v[kk]])
PY
