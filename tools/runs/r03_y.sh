#!/bin/bash
# bordered LDS glimpse copies + batched per-step LDS reads in the canvas kernels: parity first, then the probes
O=gpurun_out/r03_y; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "canvas or st_ or write or unroll" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python tools/probes/canvas_scaling.py 2>&1 | grep -v amdgpu.ids | tee $O/canvas_scaling.txt | grep -E " 64 | 1024 |scale"
echo "--- fused launch with 1024 threads"
for S in "0.45 0.65" "0.9 1.0"; do timeout 120 tools/kbench/bin/st_trace 64 3 50 20 1 4 $S 2>&1 | grep -A13 -E "^canvas_unroll_bwd\(recompute|^canvas_fused|^canvas_unroll_fwd_banded"; done > $O/st_trace.txt
grep -E "us/launch" $O/st_trace.txt
for C in c2 c4 c5 c2; do python bench.py --config $C --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', d['value'], d['ms_per_step'])"; done | tee $O/bench.txt
