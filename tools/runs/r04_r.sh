#!/bin/bash
# canvas row split at configs[3] (100x100 / 28x28 / T=5, batch 64): 320 backward units with footprints up to 60x60
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04_r
for S in 1 2 3 4 1 2; do
  echo "== AIR_CANVAS_SPLIT=$S"
  AIR_CANVAS_SPLIT=$S python bench.py --config c4 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r04_r/c4_split.txt
