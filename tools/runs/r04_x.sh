#!/bin/bash
# configs[3] (100x100 / 28x28 / T=5, batch 64): the fused canvas launch (recompute backward, T^2 taps) against the two launches
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04_x
for V in 1 0 1 0; do
  echo "== AIR_FUSE_CANVAS=$V"
  AIR_FUSE_CANVAS=$V python bench.py --config c4 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('kernel_launches_per_step'))"
done 2>&1 | tee gpurun_out/r04_x/c4_fuse.txt
