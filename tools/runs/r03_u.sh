#!/bin/bash
# wide 64x64-tile TN kernel for the short-K (K = batch) weight-gradient launch of the latency regime: A/B at c2 and c4, B = 64
O=gpurun_out/r03_u; mkdir -p $O
for MK in 256 64 256 64; do
  for C in c2 c4; do
    AIR_GEMM_WIDE_TN_MINK=$MK python bench.py --config $C --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C mink=$MK', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab.txt
  done
done
AIR_GEMM_WIDE_TN_MINK=64 timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q -k "oracle" 2>&1 | tail -2
