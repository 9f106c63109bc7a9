#!/bin/bash
O=gpurun_out/r03_p; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q -k "data_parallel" > $O/dp.log 2>&1; tail -8 $O/dp.log
