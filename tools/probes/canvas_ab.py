#!/usr/bin/env python
"""Developer probe: the row-streaming canvas kernels of round 4 (the library in the tree) against the round-3 kernels (a copy of the
old library, tools/kbench/bin/libair_hip_r03.so) on the same inputs: forward (banded), stored-canvas backward, recompute backward and
the fused launch, over a batch sweep and three scale ranges of `where`.  us per launch (HIP events, back to back)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H
from bench import event_time_ms

new = H.lib()
old_name = os.environ.get("OLD_LIB", "libair_hip_r03.so")      # libair_hip_prev.so: a build of the previous commit (ABI 6: the fused launch takes n_split)
old_abi6 = old_name != "libair_hip_r03.so"
old_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kbench", "bin", old_name)
old = ctypes.CDLL(old_path) if os.path.exists(old_path) else None
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
if old is not None:
    old.air_canvas_unroll_fwd_banded.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, F, P]
    old.air_canvas_unroll_bwd.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, P]
    old.air_canvas_unroll_fwd_bwd.argtypes = [P, P, P, P, P, P, P, I, P, P] + ([I] if old_abi6 else []) + [I, I, I, I, I, I, F, F, F, P]
    for f in (old.air_canvas_unroll_fwd_banded, old.air_canvas_unroll_bwd, old.air_canvas_unroll_fwd_bwd):
        f.restype = I
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
p = H._p
T, Hh, Ww, h, w = (int(x) for x in os.environ.get("SHAPE", "3,50,50,20,20").split(","))
HW, hw = Hh * Ww, h * w
NS = int(os.environ.get("NS", "2"))
batches = [int(x) for x in os.environ.get("BATCHES", "8,64,256,1024,8192,65536").split(",")]
print(f"T={T} canvas {Hh}x{Ww} glimpse {h}x{w}; fused launch with n_split={NS}")
print(f"{'scale':>10s} {'B':>6s} | {'fwd old':>8s} {'new':>8s} | {'bwd old':>8s} {'new':>8s} | {'rc old':>8s} {'new':>8s} | {'fused old':>9s} {'new':>8s}")
for lo, hi in ((0.2, 0.3), (0.45, 0.65), (0.9, 1.0)):
    for B in batches:
        n = T * B
        g = torch.Generator(device=dev).manual_seed(B)
        glm = torch.randn(n, hw, device=dev, generator=g)
        where = torch.empty(n, 4, device=dev)
        where[:, 0] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g); where[:, 2] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g)
        where[:, 1] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3
        pres = (torch.rand(n, device=dev, generator=g) < 0.7).float()
        obs = torch.rand(B, HW, device=dev, generator=g)
        nb = int(new.air_canvas_unroll_bands(B, Hh))
        torch.cuda.synchronize()               # the inputs are written on torch's stream, the launches go to `stream`
        out = {}
        for name, lib in (("old", old), ("new", new)):
            if lib is None:
                out[name] = [float("nan")] * 4
                continue
            steps = torch.empty(T, B, HW, device=dev); final = torch.empty(B, HW, device=dev)
            parts = torch.empty(nb, B, device=dev); dgl = torch.empty(n, hw, device=dev); dwh = torch.empty(4 * n, 4, device=dev)
            f = lambda: lib.air_canvas_unroll_fwd_banded(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, T, B, Hh, Ww, h, w, 1.0, 0.3, sp)
            bst = lambda: lib.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), p(final), p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
            brc = lambda: lib.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), None, p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
            if name == "old" and not old_abi6:
                fu = lambda: lib.air_canvas_unroll_fwd_bwd(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
            else:
                fu = lambda: lib.air_canvas_unroll_fwd_bwd(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, p(dgl), p(dwh), NS, T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
            r = []
            for fn in (f, bst, brc, fu):
                rc = fn()
                r.append(event_time_ms(lib if name == "new" else new, sp, fn, 10 if B >= 16384 else 200) * 1e3 if rc == 0 else float("nan"))
            out[name] = r
            del steps, final, parts, dgl, dwh
        o, nw = out["old"], out["new"]
        print(f"{lo:.2f}-{hi:.2f} {B:6d} | {o[0]:8.2f} {nw[0]:8.2f} | {o[1]:8.2f} {nw[1]:8.2f} | {o[2]:8.2f} {nw[2]:8.2f} | {o[3]:9.2f} {nw[3]:8.2f}")
