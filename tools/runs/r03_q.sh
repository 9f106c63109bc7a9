#!/bin/bash
# bias-gradient tiles of the bf16 weight-gradient launches: B once as fp32 instead of mirror + fp32 (A/B at batch 1024, then parity)
O=gpurun_out/r03_q2; mkdir -p $O
for V in 1 0 1 0; do AIR_GEMM_COLSUM_FROM_F32=$V python bench.py --config c5 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 colsum_from_f32=$V', d['value'], d['ms_per_step'])"; done | tee $O/ab.txt
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_engine.py -m gpu -x -q -k "bf16 or mirror or large_batch or c5 or data_path" 2>&1 | grep -E "passed|failed|Error" | tail -3
