#!/bin/bash
# round-3 GPU session F: LSTM on the bf16 data path + per-problem mirrors in the weight-gradient groups: tests and A/B benches
O=gpurun_out/r03_f; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "bf16 or lstm or shadow" > $O/kernel_tests.log 2>&1; tail -4 $O/kernel_tests.log
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q -k "bf16 or two_lane" > $O/engine_tests.log 2>&1; tail -4 $O/engine_tests.log
B="python bench.py --no-cpu-baseline --no-sweep --config c5 --steps 1000 --warmup 100"
$B > $O/bench_c5.json 2> $O/bench.log
AIR_BF16_LSTM=0 $B > $O/bench_c5_no_lstm16.json 2>> $O/bench.log
AIR_BF16_STORAGE=0 $B > $O/bench_c5_round_only.json 2>> $O/bench.log
$B > $O/bench_c5_2.json 2>> $O/bench.log
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config']['kernel_launches_per_step'], d['roofline_gemm']['achieved'], d['roofline_gemm']['gemm_us_per_step_isolated'])
except Exception as e: print('$f', 'FAILED', e)"; done
grep -v amdgpu.ids $O/bench.log | tail -5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/$O/trace -o b -- python $R/bench.py --config c5 --no-cpu-baseline --no-sweep --steps 200 --warmup 20 > /dev/null 2>> $R/$O/trace.log
python $R/tools/rocpd_summary.py $(find $R/$O/trace -name "*.db" | head -1) --by-position step_epilogue_kernel > $R/$O/positions_c5.txt
rm -rf $R/$O/trace
cat $R/$O/positions_c5.txt
