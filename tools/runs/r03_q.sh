#!/bin/bash
# f1 evidence: 20 more seeds of the 300 k-update run on the procedural-digit multi-MNIST
O=gpurun_out/r03_q; mkdir -p $O
for SEED in $(seq 13 32); do
  timeout 600 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 20000 --save-every 1000000 \
      --eval-batches 10 --summary-every 0 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/glyphs_300k_seed${SEED}_log.jsonl
  grep "Data test" $O/train_seed$SEED.log | tail -1 | cut -c1-120
done
rm -rf $O/run
