ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "mlp_dx_chain" > $OUT/chain_tests.log 2>&1; tail -15 $OUT/chain_tests.log | grep -E "passed|failed|Error|assert" | head
timeout 900 python -m pytest tests/test_engine.py -q -m gpu -k "bf16 or permutation" > $OUT/engine_bf16.log 2>&1; tail -3 $OUT/engine_bf16.log; grep -E "^(FAILED|ERROR)" $OUT/engine_bf16.log | head
PYTHONPATH=$ROOT python tools/probes/plan_dump.py c5 > $OUT/plan_c5.txt 2>&1; grep -n "chain\|^2[0-9]\|^3[0-5]" $OUT/plan_c5.txt | head -30
for ch in 0 1 0 1; do for R in 64 32; do
  AIR_DX_CHAIN_ROWS=$R AIR_DX_CHAIN=$ch timeout 300 python bench.py --config c5 --fixed-batch --no-cpu-baseline --no-sweep --no-other-configs --steps 1000 --warmup 100 2>/dev/null > $OUT/line.json
  python -c "import sys, json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('AIR_DX_CHAIN=$ch rows $R', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'])"
done; done
