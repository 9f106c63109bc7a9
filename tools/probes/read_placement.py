#!/usr/bin/env python
"""Developer probe: how much does WHERE the output buffer lands move the out-of-cache ST read?  Same images, same `where`, the output at
different byte offsets inside one large allocation (and the images likewise)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H
from bench import event_time_ms, HBM_PEAK_GBS
lib = H.lib()
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(stream.cuda_stream); p = H._p
(Hh, Ww), (h, w), T, B = (50, 50), (20, 20), 3, 65536
n = T * B
slack = 64 << 20
pool_img = torch.empty(B * Hh * Ww + slack // 4, device=dev)
pool_out = torch.empty(n * h * w + slack // 4, device=dev)
where = torch.empty(n, 4, device=dev)
where[:, 0] = 0.45 + 0.2 * torch.rand(n, device=dev); where[:, 2] = 0.45 + 0.2 * torch.rand(n, device=dev)
where[:, 1] = 0.6 * torch.rand(n, device=dev) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev) - 0.3
pool_img.uniform_(0, 1)
minimal = 4 * (B * Hh * Ww + n * (h * w + 4))
print(f"pool_img at {pool_img.data_ptr():#x}, pool_out at {pool_out.data_ptr():#x}, where at {where.data_ptr():#x}")
offs = [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 17 << 20, 32 << 20, (33 << 20) + 4096]
for which in ("out", "img"):
    for off in offs:
        o_img = pool_img[(off // 4 if which == "img" else 0):][: B * Hh * Ww].view(B, Hh, Ww)
        o_out = pool_out[(off // 4 if which == "out" else 0):][: n * h * w].view(n, h, w)
        torch.cuda.synchronize()
        fn = lambda: lib.air_st_read_fwd(p(o_img), p(where), p(o_out), n, B, Hh, Ww, h, w, sp)
        assert fn() == 0
        t = [event_time_ms(lib, sp, fn, 20) * 1e3 for _ in range(3)]
        print(f"{which} offset {off:>10d} B: " + " ".join(f"{x:7.2f}" for x in t) + f" us  ({minimal / (min(t) * 1e-6) / 1e9 / HBM_PEAK_GBS:.3f})")
