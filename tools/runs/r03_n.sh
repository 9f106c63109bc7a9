#!/bin/bash
# final round-3 evidence, part 2: unprofiled bench lines of every named configuration (PMC JSONs of this build are in profiles/)
O=gpurun_out/r03_n; mkdir -p $O
bash tools/profile_round.sh r03_n bench > $O/profile_round.log 2>&1
python bench.py --no-cpu-baseline --no-sweep --steps-per-replay 4 > $O/r03_n_bench_c2_b64_4steps_per_replay.json 2>> $O/bench.log
for f in $O/r03_n_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_other_kernels']['canvas_unroll_bwd'].get('traffic'))
except Exception as e: print('$f', 'FAILED', e)"; done
