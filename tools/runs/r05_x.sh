# Round 5, final tree (streaming short-K weight gradients in the build): whole GPU suite first, then the profiler passes and the
# unprofiled lines of the SAME binary (tag r05_t), each step bounded and skipped once the session's deadline has passed so that
# the call always ends by itself.    bash tools/runs/r05_x.sh [tag] [deadline seconds]
ROOT=$(pwd); TAG=${1:-r05_t}; DEADLINE=${2:-1380}; T0=$(date +%s)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
left() { echo $(( DEADLINE - $(date +%s) + T0 )); }
step() { local NEED=$1; shift; if [ $(left) -lt $NEED ]; then echo "SKIPPED ($(left) s left < $NEED): $*" | tee -a $OUT/session.log; return 1; fi; echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/session.log; return 0; }

step 420 "GPU suite" && { timeout 700 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/gpu_tests.log | head; }
step 60 "smoke" && { timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log; }
step 400 "profiler passes (pmc mode)" && { EVERY=2 timeout $(( $(left) - 240 > 900 ? 900 : $(left) - 240 )) bash tools/profile_round.sh $TAG pmc > $OUT/pmc.log 2>&1; rm -rf $OUT/trace* $OUT/pmc_*/; cp $OUT/${TAG}_*_instep_pmc.json $OUT/${TAG}_*_instep_durations.json profiles/ 2>/dev/null; ls $OUT | head -40; }
cd $ROOT
PY="python $ROOT/bench.py"
step 100 "default line" && timeout 200 $PY > $OUT/${TAG}_bench_c2_b64_unprofiled.json 2> $OUT/bench.log
step 50 "driver command" && timeout 150 $PY --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_c2_b64_driver_command.json 2>> $OUT/bench.log
step 40 "c4" && timeout 120 $PY --config c4 --no-cpu-baseline --steps 1000 --warmup 100 > $OUT/${TAG}_bench_c4_b64.json 2>> $OUT/bench.log
step 40 "c5" && timeout 120 $PY --config c5 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > $OUT/${TAG}_bench_c5_b1024_bf16.json 2>> $OUT/bench.log
step 40 "b1024 f32" && timeout 120 $PY --batch 1024 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > $OUT/${TAG}_bench_c2_b1024_f32.json 2>> $OUT/bench.log
python - <<PY
import json
for n in ("c2_b64_unprofiled", "c2_b64_driver_command", "c4_b64", "c5_b1024_bf16", "c2_b1024_f32"):
    try:
        d = json.loads(open("$OUT/${TAG}_bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("traffic"), (d.get("other_configs") or {}).keys())
    except Exception as e:
        print(n, "FAILED", e)
PY
echo "[$(( $(date +%s) - T0 )) s] done" | tee -a $OUT/session.log
