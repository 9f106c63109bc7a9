#!/bin/bash
# canvas backward with the contraction weights formed once per glimpse column / row: parity, then A/B against the previous build
O=gpurun_out/r04_w; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_golden.py tests/test_extreme_scales.py -q -m gpu -x -k "canvas or write or golden or degenerate or extreme" > $O/canvas_tests.log 2>&1; echo "canvas tests rc=$?"; tail -3 $O/canvas_tests.log
OLD_LIB=libair_hip_prev.so BATCHES=${BATCHES:-8,64,1024,8192,65536} python tools/probes/canvas_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/canvas_ab.txt
