#!/bin/bash
# fused `what` head (Linear + sampling + KL rows + baseline latent columns in one launch): parity, then A/B in the step
O=gpurun_out/r03_wh; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "what_head" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 900 python -m pytest tests/test_engine.py tests/test_api.py tests/test_abi_exports.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for V in 1 0 1 0; do for C in c2 c4; do AIR_FUSE_WHAT_HEAD=$V timeout 120 python bench.py --config $C --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C fuse_what_head=$V', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])"; done; done | tee $O/ab.txt
