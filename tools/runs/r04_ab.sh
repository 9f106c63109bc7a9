#!/bin/bash
# canvas forward / stored backward in the throughput regime: grid cap (AIR_CANVAS_GRID; 0 = resident workgroups from the occupancy query)
O=gpurun_out/r04_ab; mkdir -p $O
for G in 0 768 1024 1280 1536 1792 2048 3072 0; do
  echo "== AIR_CANVAS_GRID=$G"
  AIR_CANVAS_GRID=$G OLD_LIB=libair_hip_prev.so BATCHES=8192,65536 python tools/probes/canvas_ab.py 2>&1 | grep "0.45-0.65"
done | tee $O/canvas_grid.txt
