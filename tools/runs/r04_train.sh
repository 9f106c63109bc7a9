#!/bin/bash
# round 4: f1 / f3 evidence on the round's build and the reference-order dataset generator: a handful of seeds x 300 k updates of the
# training-script counterpart (procedural digits, HBM feeder), the progress figure of the best one kept (evaluation.py:31-65)
O=gpurun_out/r04_t; mkdir -p $O
for SEED in ${SEEDS:-6 14 2 10 17 31}; do
  timeout 400 python -m attend_infer_repeat_amd.scripts.multi_mnist --glyphs --iters 300000 --device-feeder --log-every 10000 --save-every 300000 \
      --figures --eval-batches 20 --seed $SEED --results-dir $O/run --run-name glyphs_seed$SEED > $O/train_seed$SEED.log 2>&1
  cp $O/run/glyphs_seed$SEED/log.jsonl $O/r04_train_glyphs_300k_seed${SEED}_log.jsonl 2>/dev/null
  python - <<PY
import json, os
from PIL import Image
seed = $SEED
rows = [json.loads(l) for l in open("$O/run/glyphs_seed%d/log.jsonl" % seed) if l.strip()]
test = [r for r in rows if r.get("data") == "test"]
last = test[-1] if test else {}
print("seed", seed, "final acc", last.get("num_step_acc"), "num_step", last.get("num_step"), "loss", last.get("loss"))
fig = "$O/run/glyphs_seed%d/progress_fig_300000.png" % seed
if os.path.exists(fig):
    im = Image.open(fig).convert("RGB")
    im = im.resize((im.width // 2, im.height // 2))
    im.save("$O/progress_fig_300000_seed%d.jpg" % seed, quality=85)
PY
  rm -rf $O/run
done
ls -la $O | head -30
