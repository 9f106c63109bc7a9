#!/usr/bin/env python
"""Same-box A/B of the stored-canvas backward: unit-major (AIR_CANVAS_BWD_IMG=0) against image-major (=1, round 5), bench.py's own sweep."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda", 0)
out = {}
for name, kw, T, batches in (("c2 50x50/20x20 T=3", {}, 3, [1024, 2048, 8192, 65536]),):
    cfg = EngineConfig(**kw)
    for rep in range(2):
        for mode in ("0", "1", "128", "512"):
            os.environ["AIR_CANVAS_BWD_IMG"] = "0" if mode == "0" else "1"
            os.environ["AIR_CANVAS_IMG_THREADS"] = mode if len(mode) > 1 else "256"
            f, b, pair = bench.canvas_write_sweep(cfg, T, batches, dev)
            for rb in b:
                out.setdefault(name, {}).setdefault(rb["batch"], {}).setdefault({"0": "unit", "1": "img"}.get(mode, "img" + mode), []).append((rb["us_per_launch"], rb["frac"]))
for name, d in out.items():
    print(name)
    for B, r in d.items():
        print("  batch %6d " % B + "  ".join("%s %s" % (k, [x[0] for x in v]) for k, v in r.items()))
