#!/bin/bash
# per-image fused canvas kernel: tests, A/B at batch 1024 (fp32 / bf16) and 256, sweep
O=gpurun_out/r03_j; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "canvas" > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q -k "large_batch or b1024 or bf16_path or throughput or permutation or enc512" > $O/engine_tests.log 2>&1; tail -3 $O/engine_tests.log
B="python bench.py --no-cpu-baseline --no-sweep --steps 1000 --warmup 100"
for C in "--config c5" "--batch 1024" "--batch 256" "--batch 4096"; do
  N=$(echo $C | tr -d ' -')
  $B $C > $O/bench_${N}_image.json 2>> $O/bench.log
  AIR_FUSE_CANVAS_IMAGE=0 $B $C > $O/bench_${N}_two_launch.json 2>> $O/bench.log
done
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
python - <<'PY'
import sys, ctypes, torch, json
sys.path.insert(0, ".")
import bench
from attend_infer_repeat_amd.engine import EngineConfig
f, b, i = bench.canvas_write_sweep(EngineConfig(), 3, [1024, 8192, 65536], torch.device("cuda", 0))
for name, rows in (("fwd", f), ("bwd", b), ("image", i)):
    print(name, [(r["batch"], r["us_per_launch"], r["frac"]) for r in rows])
PY
grep -v amdgpu.ids $O/bench.log | tail -5
