#!/bin/bash
# round 5: f1 statistics with and without the stability switch: SEEDS x 300 k updates of the training-script
# counterpart (procedural digits through the reference's generator, HBM feeder).  GUARD=1e-6 -> --guard-degenerate 1e-6; GUARD=0 -> off.
G=${GUARD:-0}; TAG=${TAG:-r05_t}
O=gpurun_out/$TAG; mkdir -p $O
for SEED in ${SEEDS:-0 1 2 3 4 5 6 7 8 9 10 11}; do
  timeout 400 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 10000 --save-every 300000 \
      --eval-batches 20 --seed $SEED --guard-degenerate $G --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/${TAG}_train_glyphs_300k_seed${SEED}_log.jsonl 2>/dev/null
  tail -1 $O/train_seed$SEED.log | cut -c1-160
  rm -rf $O/run
done
python tools/summarize_runs.py $O/${TAG}_train_glyphs_300k_seed*_log.jsonl > $O/${TAG}_summary.txt; cat $O/${TAG}_summary.txt
