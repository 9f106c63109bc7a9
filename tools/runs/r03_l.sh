#!/bin/bash
# short-K 32x32-tile groups without the intra-workgroup K split: A/B on c2 / c4
O=gpurun_out/r03_l; mkdir -p $O
AIR_GEMM_SMALLK_MAXK=256 timeout 600 python -m pytest tests/test_engine.py tests/test_hip_kernels.py -m gpu -x -q -k "forward_and_gradients or gemm" > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-sweep"
for K in 0 64 192 256; do
  AIR_GEMM_SMALLK_MAXK=$K $B > $O/bench_c2_smallk$K.json 2>> $O/bench.log
  AIR_GEMM_SMALLK_MAXK=$K $B --config c4 --steps 1000 --warmup 100 > $O/bench_c4_smallk$K.json 2>> $O/bench.log
done
$B --steps-per-replay 4 > $O/bench_c2_4steps_per_replay.json 2>> $O/bench.log
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
grep -v amdgpu.ids $O/bench.log | tail -3
