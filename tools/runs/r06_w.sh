OUT=gpurun_out/r06_w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_extreme_scales.py -q -m gpu -k "canvas or st_write or extreme or grid_stride" 2>&1 | tail -2
timeout 600 python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda", 0)
for name, kw, T in (("c4", dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5), ("c2", {}, 3)):
    cfg = EngineConfig(**kw)
    for rep in range(2):
        f, b, pair = bench.canvas_write_sweep(cfg, T, [1024, 8192, 65536], dev)
        print(name, "bwd", [(x["batch"], x["us_per_launch"], x["frac"]) for x in b], "pair", [x["frac"] for x in pair])
PY
for i in 1 2; do python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-sweep --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("c2", d['ms_per_step'], d['value'])"; done
for i in 1 2; do python bench.py --config c4 --steps 2000 --warmup 200 --no-cpu-baseline --no-sweep --no-other-configs 2>/dev/null | tail -1 | cut -c1-200; done
ONLY="canvas_unroll_bwd" timeout 120 tools/kbench/bin/st_trace_tr 2048 5 100 28 | grep -v amdgpu.ids | head -16
