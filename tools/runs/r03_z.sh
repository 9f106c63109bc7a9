#!/bin/bash
# BPTT link on 16 x 4-unit tiles (256 workgroups at batch 64): parity, then A/B in the step
O=gpurun_out/r03_z; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "lstm" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for NU in 64 0 64 0; do
  for C in c2 c4; do AIR_LSTM_BWD_NU4_MAX_TILES=$NU python bench.py --config $C --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C nu4_max=$NU', d['value'], d['ms_per_step'])"; done
done | tee $O/ab.txt
