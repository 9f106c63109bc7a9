# entry fold with the fast tanh: kernel parity + A/B against AIR_LSTM_BWD_ENTRY=0 + positions
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_o; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_engine.py -x -q -m gpu -k "lstm_bwd_entry or forward_and_gradients or updates_match or graph_replay" > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/tests.log | head -20
for i in 1 2 3; do
  for V in 0 1; do
    AIR_LSTM_BWD_ENTRY=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/c2_e${V}_$i.json
    AIR_LSTM_BWD_ENTRY=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_e${V}_$i.json
    python - <<PY
import json
for c in ("c2", "c4"):
    d=json.load(open("$OUT/%s_e${V}_$i.json" % c)); print(c, "entry=$V run $i", d["ms_per_step"], d["value"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --no-other-configs --steps 1500 --warmup 100 > $OUT/profiled.json 2> $OUT/profiled.log
python $ROOT/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-position gemm_grouped_opt_kernel --every 2 > $OUT/positions_c2.txt 2>&1
rm -rf $OUT/trace
cut -c1-100 $OUT/positions_c2.txt | sed -n 24,32p
