#!/bin/bash
# the round's last full GPU suite (with the seeded sweeps), parity margins and smoke
O=gpurun_out/r03_t2; mkdir -p $O
AIR_PARITY_MARGINS=$PWD/$O/r03_parity_margins.json timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; grep -E "passed|failed" $O/r03_gpu_tests.log | tail -1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
