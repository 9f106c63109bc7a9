#!/bin/bash
# f1 evidence on the round's final build: seeds 3..32 x 300 k updates (seeds 0-2 ran in r03_final2.sh), procedural digits, device feeder
O=gpurun_out/r03_s; mkdir -p $O
for SEED in $(seq 3 32); do
  timeout 600 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 10000 --save-every 1000000 \
      --eval-batches 20 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/final_build_glyphs_300k_seed${SEED}_log.jsonl 2>/dev/null
  rm -rf $O/run
  tail -2 $O/train_seed$SEED.log | grep -o "Step 300000.*num_step = [0-9.]*" | cut -c1-120
done
