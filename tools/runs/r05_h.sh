# Round 5 (late): the new engine switches (non-analytic prior, priors at None, continuous steps) under the GPU tests, then a same-box
# A/B of the LSTM weight gradients' placement (AIR_LSTM_DW_EARLY=0/1, alternating, unprofiled headline lines) and its positions.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_h; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_engine.py tests/test_api.py tests/test_abi_exports.py -x -q -m gpu \
  -k "switches or folded_closing or argument_combinations or abi or riders or graph_replay or updates_match" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for i in 1 2 3; do
  for V in 0 1; do
    AIR_LSTM_DW_EARLY=$V timeout 300 python bench.py --no-other-configs --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/bench_early${V}_$i.json
    python - <<PY
import json; d=json.load(open("$OUT/bench_early${V}_$i.json")); print("early=$V run $i", d["ms_per_step"], d["value"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  AIR_LSTM_DW_EARLY=$V timeout 600 rocprofv3 --kernel-trace -d $OUT/trace_$V -o b -- python $ROOT/bench.py --no-other-configs --steps 1500 --warmup 100 > $OUT/profiled_$V.json 2> $OUT/profiled_$V.log
  DB=$(find $OUT/trace_$V -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB --by-position gemm_grouped_opt_kernel $([ $V = 1 ] && echo "--every 2") > $OUT/positions_early$V.txt 2>&1
  rm -rf $OUT/trace_$V
  tail -6 $OUT/positions_early$V.txt | cut -c1-110
done
