#!/usr/bin/env python
"""Developer probe: the canvas kernels in the throughput regime, launched back to back, for rocprofv3 --pmc passes
(which instruction classes the launch spends its issue slots on).  B, T, scale range from the environment."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H

L = H.lib()
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
p = H._p
T, Hh, Ww, h, w = (int(x) for x in os.environ.get("SHAPE", "3,50,50,20,20").split(","))
B = int(os.environ.get("B", "8192")); lo, hi = (float(x) for x in os.environ.get("SCALE", "0.45,0.65").split(","))
REPS = int(os.environ.get("REPS", "20"))
HW, hw, n = Hh * Ww, h * w, T * B
g = torch.Generator(device=dev).manual_seed(B)
glm = torch.randn(n, hw, device=dev, generator=g)
where = torch.empty(n, 4, device=dev)
where[:, 0] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g); where[:, 2] = lo + (hi - lo) * torch.rand(n, device=dev, generator=g)
where[:, 1] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev, generator=g) - 0.3
pres = (torch.rand(n, device=dev, generator=g) < 0.7).float()
obs = torch.rand(B, HW, device=dev, generator=g)
nb = int(L.air_canvas_unroll_bands(B, Hh))
steps = torch.empty(T, B, HW, device=dev); final = torch.empty(B, HW, device=dev)
parts = torch.empty(nb, B, device=dev); dgl = torch.empty(n, hw, device=dev); dwh = torch.empty(4 * n, 4, device=dev)
gl_out = torch.empty(n, hw, device=dev)
for _ in range(REPS):
    assert L.air_canvas_unroll_fwd_banded(p(glm), p(where), p(pres), p(obs), p(steps), p(final), p(parts), nb, T, B, Hh, Ww, h, w, 1.0, 0.3, sp) == 0
    assert L.air_canvas_unroll_bwd(p(glm), p(where), p(pres), p(obs), p(final), p(dgl), p(dwh), T, B, Hh, Ww, h, w, 1.0, 0.3, 1.0 / B, sp) == 0
    assert L.air_st_read_fwd(p(obs), p(where), p(gl_out), n, B, Hh, Ww, h, w, sp) == 0
torch.cuda.synchronize()
print("done", B, T, lo, hi)
