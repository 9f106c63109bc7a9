#!/bin/bash
# final round-3 evidence, part 1: full GPU suite with parity margins + PMC / traces of the final build
O=gpurun_out/r03_m; mkdir -p $O
AIR_PARITY_MARGINS=$PWD/$O/r03_parity_margins.json timeout 1800 python -m pytest tests -m gpu -q > $O/r03_gpu_tests.log 2>&1; tail -4 $O/r03_gpu_tests.log
bash tools/profile_round.sh r03_m pmc > $O/profile_round.log 2>&1; ls $O | head -40
