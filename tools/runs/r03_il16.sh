#!/bin/bash
# 16-byte interleaved bf16 operand loads (lane-pair exchange): parity, then A/B at batch 1024
O=gpurun_out/r03_il16; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine.py -m gpu -x -q -k "bf16 or mirror or large_batch or data_path or gemm" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for V in 1 0 1 0; do AIR_GEMM_BF16_IL16=$V python bench.py --config c5 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 il16=$V', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
