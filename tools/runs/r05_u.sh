# final tree: the default bench line (graph-node event timer, all-cores replica leg) first, then the whole GPU suite
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
AIR_BENCH_STREAM_TIMER=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $OUT/bench_stream_timer.json 2>> $OUT/bench_default.log
python - <<PY
import json
for n in ("default", "stream_timer"):
    try:
        d = json.loads(open("$OUT/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "live", d["roofline"].get("frac_live"), d["roofline"].get("us_per_launch_live_back_to_back"),
              "gemm", d["roofline_gemm"]["gemm_us_per_step_isolated"], [(e["batch"], e["us_per_launch"]) for e in d["roofline_sweep_st_read_fwd"]][:3],
              [(e["batch"], e["us_per_launch"]) for e in d["roofline_sweep_canvas_write_bwd"]][:2], (d.get("cpu_baseline") or {}).get("all_host_cores_replicas"))
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -3 $OUT/bench_default.log
timeout 1700 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/gpu_tests.log 2>&1
grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -2; grep -A14 "slowest" $OUT/gpu_tests.log | head -16
