#!/bin/bash
# round-3 GPU session H: the unprofiled bench lines of every named configuration (PMC JSONs of the same build are in profiles/),
# plus the per-position trace of the c2/b64 step (one step per period)
O=gpurun_out/r03_h; mkdir -p $O
bash tools/profile_round.sh r03_h bench > $O/profile_round.log 2>&1
for f in $O/r03_h_bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_other_kernels']['canvas_unroll_bwd'].get('traffic'))
except Exception as e: print('$f', 'FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/$O/trace -o b -- python $R/bench.py --no-cpu-baseline --no-sweep --steps 600 --warmup 60 > /dev/null 2>> $R/$O/trace.log
python $R/tools/rocpd_summary.py $(find $R/$O/trace -name "*.db" | head -1) --by-position step_epilogue_kernel > $R/$O/r03_h_positions_c2_b64.txt
rm -rf $R/$O/trace
head -40 $R/$O/r03_h_positions_c2_b64.txt
