#!/bin/bash
# final build: parity margins of the GPU suite, the learning-dynamics test with its report, and three 300 k-update runs (seeds 0-2)
O=gpurun_out/r03_r; mkdir -p $O
AIR_PARITY_MARGINS=$PWD/$O/r03_parity_margins.json AIR_DYNAMICS_REPORT=$PWD/$O/dynamics_report.json timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r03_gpu_tests.log | tail -1
for SEED in 0 1 2; do
  timeout 1200 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 10000 --save-every 100000 \
      --eval-batches 20 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/final_build_glyphs_300k_seed${SEED}_log.jsonl
  tail -2 $O/train_seed$SEED.log
done
rm -rf $O/run
