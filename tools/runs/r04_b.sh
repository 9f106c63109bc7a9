#!/bin/bash
# round 4, call b: seed 9's early non-finite update with per-buffer diagnostics; the extreme-scale tests
O=gpurun_out/r04_b; mkdir -p $O
timeout 900 python tools/blowup_replay.py --seed 9 --out $O/blowup > $O/blowup_seed9.log 2>&1; echo "seed 9 rc=$?"; tail -12 $O/blowup_seed9.log
rm -f $O/blowup/*.pt
timeout 900 python -m pytest tests/test_extreme_scales.py -q -m gpu > $O/extreme.log 2>&1; echo "extreme rc=$?"; tail -40 $O/extreme.log
