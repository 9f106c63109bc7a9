mkdir -p gpurun_out/r03_b; O=gpurun_out/r03_b
./tools/kbench/bin/graph_branch 300 64 > $O/graph_branch_300_64.txt 2>&1
./tools/kbench/bin/graph_branch 400 192 > $O/graph_branch_400_192.txt 2>&1
cat $O/graph_branch_300_64.txt
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q -k "two_lane or riders or replay or graph_captured" > $O/two_lane_tests.log 2>&1; tail -5 $O/two_lane_tests.log
python bench.py --no-cpu-baseline --no-sweep > $O/bench_two_lane.json 2> $O/bench.log
AIR_TWO_LANE=0 python bench.py --no-cpu-baseline --no-sweep > $O/bench_linear.json 2>> $O/bench.log
HIP_FORCE_DEV_KERNARG=0 python bench.py --no-cpu-baseline --no-sweep > $O/bench_two_lane_kernarg0.json 2>> $O/bench.log
HIP_FORCE_DEV_KERNARG=1 python bench.py --no-cpu-baseline --no-sweep > $O/bench_two_lane_kernarg1.json 2>> $O/bench.log
python bench.py --config c4 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > $O/bench_c4_two_lane.json 2>> $O/bench.log
AIR_TWO_LANE=0 python bench.py --config c4 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > $O/bench_c4_linear.json 2>> $O/bench.log
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['config'].get('kernel_launches_by_lane'))"; done
tail -5 $O/bench.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
