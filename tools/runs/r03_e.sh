#!/bin/bash
# round-3 GPU session E: fused canvas launch (A/B), new tests, more seeds of the 300 k glyph run
O=gpurun_out/r03_e; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py tests/test_golden.py tests/test_api.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
AIR_DYNAMICS_REPORT=$PWD/$O/dynamics_report.json timeout 900 python -m pytest tests/test_training_dynamics.py -m gpu -x -q > $O/dynamics_test.log 2>&1; tail -3 $O/dynamics_test.log
B="python bench.py --no-cpu-baseline --no-sweep"
$B > $O/bench_fused_canvas.json 2> $O/bench.log
AIR_FUSE_CANVAS=0 $B > $O/bench_unfused_canvas.json 2>> $O/bench.log
$B > $O/bench_fused_canvas_2.json 2>> $O/bench.log
AIR_FUSE_CANVAS=0 $B > $O/bench_unfused_canvas_2.json 2>> $O/bench.log
$B --config c4 --steps 1000 --warmup 100 > $O/bench_c4_fused.json 2>> $O/bench.log
AIR_FUSE_CANVAS=0 $B --config c4 --steps 1000 --warmup 100 > $O/bench_c4_unfused.json 2>> $O/bench.log
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
grep -v amdgpu.ids $O/bench.log | tail -5
for SEED in 3 4 5 6 7 8 9 10 11 12; do
  timeout 600 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 20000 --save-every 1000000 \
      --eval-batches 10 --summary-every 0 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/glyphs_300k_seed${SEED}_log.jsonl
  grep "Data test" $O/train_seed$SEED.log | tail -1 | cut -c1-150
done
rm -rf $O/run
