#!/bin/bash
O=gpurun_out/r03_t; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; tail -4 $O/r03_gpu_tests.log | grep -E "passed|failed"
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
