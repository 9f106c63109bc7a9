#!/usr/bin/env python
"""Developer probe: is the fused canvas launch (air_canvas_unroll_fwd_bwd) bound by the chain inside one workgroup or by the chip?
Times the launch and its two roles on their own over a batch sweep (few workgroups -> chain; many -> chip) and for three scale ranges
of `where` (the footprint a backward unit walks).  Run on the GPU box:  python tools/probes/canvas_scaling.py"""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H
from bench import event_time_ms

lib = H.lib()
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
p = H._p
T, Hh, Ww, h, w = 3, 50, 50, 20, 20
NS = int(os.environ.get("NS", "1"))          # workgroups per backward unit of the fused launch
HW, hw = Hh * Ww, h * w
print(f"{'scale':>10s} {'B':>5s} {'bands':>5s} {'fused':>8s} {'fwd':>8s} {'bwd_rc':>8s} {'bwd_st':>8s}   (us per launch)")
for lo, hi in ((0.2, 0.3), (0.45, 0.65), (0.9, 1.0)):
    for B in (2, 8, 32, 64, 128, 256, 512):
        n = T * B
        g = torch.Generator(device=dev).manual_seed(B)
        glm = torch.randn(n, hw, device=dev, generator=g)
        where = torch.empty(n, 4, device=dev)
        where[:, 0] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g); where[:, 2] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g)
        where[:, 1] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3
        pres = (torch.rand(n, device=dev, generator=g) < 0.7).float()
        obs = torch.rand(B, HW, device=dev, generator=g)
        steps = torch.empty(T, B, HW, device=dev); final = torch.empty(B, HW, device=dev)
        nb = int(lib.air_canvas_unroll_bands(B, Hh))
        parts = torch.empty(nb, B, device=dev); dgl = torch.empty(n, hw, device=dev); dwh = torch.empty(4 * n, 4, device=dev)
        torch.cuda.synchronize()
        fused = lambda: lib.air_canvas_unroll_fwd_bwd(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, p(dgl), p(dwh),
                                                      NS, T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
        f = lambda: lib.air_canvas_unroll_fwd_banded(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, T, B,
                                                     Hh, Ww, h, w, 1.0, 0.3, sp)
        brc = lambda: lib.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), None, p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
        bst = lambda: lib.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), p(final), p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp)
        r = []
        for fn in (fused, f, brc, bst):
            try:
                rc = fn()
                r.append(event_time_ms(lib, sp, fn, 200) * 1e3 if rc == 0 else float("nan"))
            except Exception:
                r.append(float("nan"))
        print(f"{lo:.2f}-{hi:.2f} {B:5d} {nb:5d} " + " ".join(f"{x:8.2f}" for x in r))
