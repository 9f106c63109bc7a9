#!/usr/bin/env python
"""Round 6: same-box A/B of the stored-canvas backward -- the pixel-pass kernels of rounds 2-5 (AIR_CANVAS_BWD_GS=0) against the
glimpse-space form (=1) -- on bench.py's own sweep at configs[1] and configs[3] shapes, at the sweep's scales (0.45-0.65) and at the
scales the slow configs[3] states sit at (1.4-2.8).  The pixel-pass kernels and the AIR_CANVAS_BWD_GS switch left the tree once the
A/Bs were recorded (profiles/r06_canvas_gs_ab.txt): on the current tree the "old" leg times the same kernel as "gs"; check out the
commit "Data-parallel tests with four and eight ranks sharing one GPU" (or any earlier round-6 commit) to repeat the comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda", 0)
modes = [("old", {"AIR_CANVAS_BWD_GS": "0"}), ("gs", {"AIR_CANVAS_BWD_GS": "1"})]
for extra in sys.argv[1:]:                      # e.g. gs512:AIR_CANVAS_GS_THREADS=512
    name, kv = extra.split(":")
    modes.append((name, dict([("AIR_CANVAS_BWD_GS", "1")] + [x.split("=") for x in kv.split(",")])))
KEYS = ("AIR_CANVAS_BWD_GS", "AIR_CANVAS_GS_THREADS", "AIR_CANVAS_BWD_IMG", "AIR_CANVAS_GRID", "AIR_CANVAS_IMG_MIN_UNITS")
cases = (("c4 100x100/28x28 T=5", dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5, [64, 1024, 8192, 65536]),
         ("c2 50x50/20x20 T=3", {}, 3, [64, 1024, 8192, 65536]))
for name, kw, T, batches in cases:
    cfg = EngineConfig(**kw)
    for sname, scale in (("scales 0.45-0.65", (0.45, 0.2)), ("scales 1.4-2.8", (1.4, 1.4))):
        out = {}
        for rep in range(2):
            for mname, env in modes:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                f, b, pair = bench.canvas_write_sweep(cfg, T, batches if scale[0] < 1 else batches[:3], dev, scale=scale)
                for rb, rp in zip(b, pair):
                    out.setdefault(rb["batch"], {}).setdefault(mname, []).append((rb["us_per_launch"], rb["frac"], rp["frac"]))
        print(name, "|", sname, "| backward us per launch [runs], frac of 8 TB/s (bwd, fwd+bwd pair)")
        for B, r in out.items():
            print("  images %6d  " % B + "   ".join("%s %s f=%.3f pair=%.3f" % (k, [x[0] for x in v], v[-1][1], v[-1][2]) for k, v in r.items()))

# the fused forward + recompute-form backward launch of the latency regime (configs[1] step position 14)
import ctypes
from attend_infer_repeat_amd import hip as H
lib = H.lib()
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
p = H._p
for name, (Hh, Ww, h, w, T), B in (("c2 fused", (50, 50, 20, 20, 3), 64), ("c4 fused", (100, 100, 28, 28, 5), 64)):
    for sname, scale in (("0.45-0.65", (0.45, 0.2)), ("1.4-2.8", (1.4, 1.4))):
        n, HW, hw = T * B, Hh * Ww, h * w
        g = torch.Generator(device=dev).manual_seed(B)
        glm = torch.randn(n, hw, device=dev, generator=g)
        where = torch.empty(n, 4, device=dev)
        where[:, 0] = scale[0] + scale[1] * torch.rand(n, device=dev, generator=g); where[:, 2] = scale[0] + scale[1] * torch.rand(n, device=dev, generator=g)
        where[:, 1] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3
        pres = (torch.rand(n, device=dev, generator=g) < 0.7).float()
        obs = torch.rand(B, HW, device=dev, generator=g)
        steps = torch.empty(T, B, HW, device=dev); final = torch.empty(B, HW, device=dev)
        nb = int(lib.air_canvas_unroll_bands(B, Hh))
        parts = torch.empty(nb, B, device=dev); dgl = torch.empty(n, hw, device=dev); dwh = torch.empty(4, n, 4, device=dev)
        res = {}
        for rep in range(2):
            for mname, env in modes[:2]:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                if lib.air_canvas_unroll_fwd_bwd_fits(nb, 1, T, B, Hh, Ww, h, w) != 1:
                    continue
                fn = lambda: lib.air_canvas_unroll_fwd_bwd(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, p(dgl), p(dwh), 1,
                                                           T, B, Hh, Ww, h, w, 0.5, 0.3, 1.0 / B, sp)
                res.setdefault(mname, []).append(round(bench.event_time_ms(lib, sp, fn, 200) * 1e3, 2))
        print(name, "batch", B, "scales", sname, "us per fused launch:", res)
