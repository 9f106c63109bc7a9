#!/bin/bash
# round-3 GPU session G: full GPU suite (with the parity margins recorded), then the PMC / trace evidence of this build
O=gpurun_out/r03_g; mkdir -p $O
AIR_PARITY_MARGINS=$PWD/$O/r03_parity_margins.json timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; tail -8 $O/r03_gpu_tests.log
bash tools/profile_round.sh r03_g pmc > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log
