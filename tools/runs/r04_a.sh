#!/bin/bash
# round 4, call a: the extreme-scale edge tests, the whole GPU suite on the autodiff-order KL gradient, and the blow-up replay
O=gpurun_out/r04_a; mkdir -p $O
timeout 900 python -m pytest tests/test_extreme_scales.py -x -q -m gpu > $O/extreme.log 2>&1; echo "extreme rc=$?"; tail -15 $O/extreme.log
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log
for SEED in 30 26 32 1 8 9 18; do
  timeout 1500 python tools/blowup_replay.py --seed $SEED --out $O/blowup > $O/blowup_seed$SEED.log 2>&1
  echo "seed $SEED rc=$?"; tail -3 $O/blowup_seed$SEED.log
  if grep -q engine_first_nonfinite_update $O/blowup/report_seed$SEED.json 2>/dev/null; then break; fi
done
rm -f $O/blowup/*.pt
