timeout 900 python -m pytest tests/test_engine.py tests/test_api.py tests/test_abi_exports.py -x -q -m gpu -k "feeder or training_script or abi or launch" > gpurun_out/r06_ai.log 2>&1; grep -E "passed|failed" gpurun_out/r06_ai.log | tail -2
B="--steps 2000 --warmup 200 --no-cpu-baseline --no-sweep --no-other-configs"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'])"; }
for i in 1 2 3; do
timeout 300 python bench.py --config c5 $B 2>/dev/null | show c5_fold
AIR_FOLD_GATHER=0 timeout 300 python bench.py --config c5 $B 2>/dev/null | show c5_nofold
done
