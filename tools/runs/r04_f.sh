#!/bin/bash
# round 4, call f: the lean throughput backward at several occupancy bounds (same process: the switch is read once per process)
O=gpurun_out/r04_f; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "canvas or st_write" > $O/canvas_tests.log 2>&1; echo "canvas tests rc=$?"; tail -4 $O/canvas_tests.log
for L in 0 5 6 7 8; do
  echo "== AIR_CANVAS_BWD_LEAN=$L"
  AIR_CANVAS_BWD_LEAN=$L BATCHES=1024,8192,65536 timeout 600 python tools/probes/canvas_ab.py 2>&1 | grep -v amdgpu.ids | awk '{print $1, $2, $7, $8}' 
done > $O/lean_sweep.txt 2>&1
cat $O/lean_sweep.txt
