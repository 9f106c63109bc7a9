#!/bin/bash
# round-3 f1 evidence: (1) engine vs CPU-oracle learning curves over 1500 updates (tests/test_training_dynamics.py, report kept);
# (2) the training-script counterpart for 300 k updates on the procedural-digit multi-MNIST, three seeds, device feeder
O=gpurun_out/r03_train; mkdir -p $O
AIR_DYNAMICS_REPORT=$PWD/$O/dynamics_report.json timeout 900 python -m pytest tests/test_training_dynamics.py -m gpu -x -q > $O/dynamics_test.log 2>&1; tail -5 $O/dynamics_test.log
for SEED in 0 1 2; do
  timeout 1200 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 10000 --save-every 100000 \
      --eval-batches 20 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/glyphs_300k_seed${SEED}_log.jsonl
  tail -4 $O/train_seed$SEED.log
done
rm -rf $O/run
