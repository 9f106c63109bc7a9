OUT=gpurun_out/r06_q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "mlp_dx_chain" 2>&1 | tail -3
timeout 120 python tools/probes/dx_chain_time.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_engine.py -q -m gpu -k "bf16 or permutation" 2>&1 | tail -4
for ch in 0 1 0 1; do
  AIR_DX_CHAIN=$ch timeout 300 python bench.py --config c5 --fixed-batch --no-cpu-baseline --no-sweep --no-other-configs --steps 1000 --warmup 100 2>/dev/null > $OUT/line.json
  python -c "import sys, json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('AIR_DX_CHAIN=$ch', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'])"
done
