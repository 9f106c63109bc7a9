#!/bin/bash
# why do good runs end in NaN?  seed 30 (0.98 accuracy at 180 k, zero steps + NaN at 200 k): diagnostics every 250 updates from 178 k
O=gpurun_out/r03_r; mkdir -p $O
timeout 900 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 200000 --device-feeder --log-every 20000 --save-every 1000000 \
    --eval-batches 10 --summary-every 0 --seed 30 --check-every 250 --check-from 178000 --results-dir $O/run --run-name s30 > $O/train.log 2>&1
cp $O/run/s30/log.jsonl $O/seed30_check_log.jsonl; rm -rf $O/run
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r03_r/seed30_check_log.jsonl") if '"check"' in l]
bad = next((i for i, r in enumerate(rows) if not all(r["finite"].values()) or r["num_step"] == 0), len(rows) - 1)
for r in rows[max(0, bad - 8):bad + 3]:
    print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items() if k not in ("data",)})
PY
