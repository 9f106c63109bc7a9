#!/bin/bash
# round 4, call d: the row-streaming canvas kernels: kernel tests, engine tests, extreme-scale tests
O=gpurun_out/r04_d; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "canvas or st_write" > $O/canvas_tests.log 2>&1; echo "canvas tests rc=$?"; tail -15 $O/canvas_tests.log
timeout 1200 python -m pytest tests/test_extreme_scales.py -q -m gpu > $O/extreme.log 2>&1; echo "extreme rc=$?"; tail -30 $O/extreme.log
timeout 1500 python -m pytest tests/test_engine.py -q -m gpu -x > $O/engine.log 2>&1; echo "engine rc=$?"; tail -15 $O/engine.log
