#!/bin/bash
# round-3 GPU session C: graph-branch probe, new kernel tests, two-lane / recompute / bf16-storage A-B benches, full GPU suite
O=gpurun_out/r03_c; mkdir -p $O
./tools/kbench/bin/graph_branch 300 64 > $O/graph_branch_300_64.txt 2>&1
./tools/kbench/bin/graph_branch 400 192 > $O/graph_branch_400_192.txt 2>&1
cat $O/graph_branch_300_64.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "bf16_data_path or shadow or recompute or rmsprop or canvas_unroll or gemm_wide" > $O/kernel_tests.log 2>&1; tail -4 $O/kernel_tests.log
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q -k "two_lane or riders or replay or graph_captured or bf16" > $O/engine_tests.log 2>&1; tail -4 $O/engine_tests.log
B="python bench.py --no-cpu-baseline --no-sweep"
$B > $O/bench_two_lane.json 2> $O/bench.log
AIR_CANVAS_RECOMPUTE=0 $B > $O/bench_two_lane_norecompute.json 2>> $O/bench.log
AIR_TWO_LANE=0 $B > $O/bench_linear.json 2>> $O/bench.log
HIP_FORCE_DEV_KERNARG=0 $B > $O/bench_two_lane_kernarg0.json 2>> $O/bench.log
$B --config c4 --steps 1000 --warmup 100 > $O/bench_c4_two_lane.json 2>> $O/bench.log
AIR_TWO_LANE=0 $B --config c4 --steps 1000 --warmup 100 > $O/bench_c4_linear.json 2>> $O/bench.log
$B --config c5 --steps 1000 --warmup 100 > $O/bench_c5_bf16_storage.json 2>> $O/bench.log
AIR_BF16_STORAGE=0 $B --config c5 --steps 1000 --warmup 100 > $O/bench_c5_bf16_round_only.json 2>> $O/bench.log
$B --batch 1024 --steps 1000 --warmup 100 > $O/bench_b1024_f32.json 2>> $O/bench.log
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config'].get('kernel_launches_by_lane'), d['roofline_gemm']['achieved'], d['roofline_gemm']['gemm_us_per_step_isolated'])
except Exception as e: print('$f', 'FAILED', e)"; done
tail -5 $O/bench.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
