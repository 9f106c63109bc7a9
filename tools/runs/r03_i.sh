#!/bin/bash
# experiment: the fused canvas launch in the throughput regime
O=gpurun_out/r03_i; mkdir -p $O
timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q -k "large_batch or b1024 or bf16_path" > $O/tests_default.log 2>&1; tail -3 $O/tests_default.log
AIR_FUSE_CANVAS_THROUGHPUT=1 timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q -k "large_batch or b1024 or bf16_path" > $O/tests_fused.log 2>&1; tail -3 $O/tests_fused.log
B="python bench.py --no-cpu-baseline --no-sweep --steps 1000 --warmup 100"
for C in "--config c5" "--batch 1024" "--batch 256"; do
  N=$(echo $C | tr -d ' -')
  $B $C > $O/bench_${N}_default.json 2>> $O/bench.log
  AIR_FUSE_CANVAS_THROUGHPUT=1 $B $C > $O/bench_${N}_fused.json 2>> $O/bench.log
done
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
grep -v amdgpu.ids $O/bench.log | tail -5
