#!/bin/bash
# per-image canvas kernel without the go buffer (7 workgroups per CU): tests + A/B by batch
O=gpurun_out/r03_k; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "canvas" > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
B="python bench.py --no-cpu-baseline --no-sweep --steps 600 --warmup 60"
for C in "--batch 1024" "--batch 2048" "--batch 4096" "--batch 8192"; do
  N=$(echo $C | tr -d ' -')
  AIR_FUSE_CANVAS_IMAGE_MIN_BATCH=1 $B $C > $O/bench_${N}_image.json 2>> $O/bench.log
  AIR_FUSE_CANVAS_IMAGE=0 $B $C > $O/bench_${N}_two_launch.json 2>> $O/bench.log
done
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from attend_infer_repeat_amd.engine import EngineConfig
f, b, i = bench.canvas_write_sweep(EngineConfig(), 3, [1024, 8192, 65536], torch.device("cuda", 0))
for name, rows in (("fwd", f), ("bwd", b), ("image", i)):
    print(name, [(r["batch"], r["us_per_launch"], r["frac"]) for r in rows])
PY
grep -v amdgpu.ids $O/bench.log | tail -3
