OUT=gpurun_out/r06_t; mkdir -p $OUT
timeout 600 python -m pytest tests/test_engine.py -q -m gpu -k "feeder" 2>&1 | tail -6
for f in 0 1 0 1; do
  AIR_FOLD_GATHER=$f timeout 300 python bench.py --no-cpu-baseline --no-sweep --no-other-configs --steps 2000 --warmup 200 2>/dev/null > $OUT/line.json
  python -c "import json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('c2 AIR_FOLD_GATHER=$f', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], 'fixed', d['fixed_batch']['ms_per_step'])"
  AIR_FOLD_GATHER=$f timeout 300 python bench.py --config c4 --no-cpu-baseline --no-sweep --no-other-configs --steps 1000 --warmup 100 2>/dev/null > $OUT/line.json
  python -c "import json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('c4 AIR_FOLD_GATHER=$f', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], 'fixed', d['fixed_batch']['ms_per_step'])"
done
for f in 0 1; do AIR_FOLD_GATHER=$f timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-other-configs 2>/dev/null > $OUT/line.json
  python -c "import json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('driver cmd AIR_FOLD_GATHER=$f', d['ms_per_step'], d['value'], d['config']['kernel_launches_per_step'], 'fixed', d['fixed_batch']['ms_per_step'])"; done
