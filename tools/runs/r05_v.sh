# Round 5 (last session): short-K weight gradients on the streaming body (AIR_GEMM_SHORTK=0/1): parity, then same-box A/B at configs[1] / [3]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_v; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine.py -x -q -m gpu -k "short_k or gemm or linear or forward_and_gradients or graph_replay or updates_match or folded or riders or launches_per" > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/tests.log | head -20
for i in 1 2; do
  for V in 0 1; do
    AIR_GEMM_SHORTK=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/c2_s${V}_$i.json
    AIR_GEMM_SHORTK=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_s${V}_$i.json
    python - <<PY
import json
for c in ("c2", "c4"):
    d=json.load(open("$OUT/%s_s${V}_$i.json" % c)); print(c, "shortk=$V run $i", d["ms_per_step"], d["value"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for C in c2 c4; do
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $ROOT/bench.py --config $C --no-cpu-baseline --no-sweep --no-other-configs --steps 1000 --warmup 100 > $OUT/profiled.json 2> $OUT/profiled.log
python $ROOT/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-position gemm_grouped_opt --every 2 > $OUT/positions_$C.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) | head -30 > $OUT/stats_$C.txt 2>&1
rm -rf $OUT/trace
cut -c1-100 $OUT/positions_$C.txt | tail -18
done
