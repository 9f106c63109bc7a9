OUT=gpurun_out/r06_s; mkdir -p $OUT
timeout 300 python -m pytest tests/test_engine.py -q -m gpu -k "feeder or gather or dataset" 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.json 2>/dev/null
timeout 600 python bench.py > $OUT/default.json 2>$OUT/default.log
python - <<PY
import json
for n in ("driver_cmd", "default"):
    d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    oc = d.get("other_configs") or {}
    print(n, d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["config"].get("model_state_at_end"), "| fixed", (d.get("fixed_batch") or {}).get("ms_per_step"), "| roofline", d["roofline"]["frac"], d["roofline"].get("duration_source"), "| gemm", d["roofline_gemm"].get("frac"), d["roofline_gemm"].get("mfma_busy_utilisation"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    for k, v in oc.items():
        print("  ", k, v.get("ms_per_step"), v.get("per_seed_spread"), [p["ms_per_step"] for p in v.get("per_seed", [])], "fixed", (v.get("fixed_batch") or {}).get("ms_per_step"), "probe", (v.get("all_steps_present_probe") or {}).get("ms_per_step"), v.get("error"))
        for kk in ("roofline_sweep_st_read_fwd", "roofline_sweep_canvas_write_fwd", "roofline_sweep_canvas_write_bwd", "roofline_sweep_canvas_write_pair"):
            if kk in v: print("     ", kk, [(x["batch"], x["frac"]) for x in v[kk]])
    for kk in ("roofline_sweep_st_read_fwd", "roofline_sweep_canvas_write_fwd", "roofline_sweep_canvas_write_bwd", "roofline_sweep_canvas_write_pair"):
        if kk in d: print("  c2 ", kk, [(x["batch"], x["frac"]) for x in d[kk]])
PY
