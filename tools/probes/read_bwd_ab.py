#!/usr/bin/env python
"""Developer probe: air_st_read_bwd (dwhere only, T glimpses per image) of the tree's library against libair_hip_prev.so, same buffers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H
from bench import event_time_ms
new = H.lib()
prev = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kbench", "bin", "libair_hip_prev.so"))
P, I = ctypes.c_void_p, ctypes.c_int
prev.air_st_read_bwd.argtypes = [P, P, P, P, P, I, I, I, I, I, I, P]; prev.air_st_read_bwd.restype = I
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(stream.cuda_stream); p = H._p
for (Hh, Ww), (h, w), T in (((50, 50), (20, 20), 3), ((100, 100), (28, 28), 5)):
    for B in (1024, 8192, 65536):
        n = T * B
        img = torch.rand(B, Hh, Ww, device=dev); dgl = torch.randn(n, h * w, device=dev)
        where = torch.empty(n, 4, device=dev)
        where[:, 0] = 0.45 + 0.2 * torch.rand(n, device=dev); where[:, 2] = 0.45 + 0.2 * torch.rand(n, device=dev)
        where[:, 1] = 0.6 * torch.rand(n, device=dev) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev) - 0.3
        d_new = torch.empty(n, 4, device=dev); d_prev = torch.empty(n, 4, device=dev)
        f_new = lambda: new.air_st_read_bwd(p(img), p(where), p(dgl), p(d_new), None, n, B, Hh, Ww, h, w, sp)
        f_prev = lambda: prev.air_st_read_bwd(p(img), p(where), p(dgl), p(d_prev), None, n, B, Hh, Ww, h, w, sp)
        torch.cuda.synchronize()
        assert f_new() == 0 and f_prev() == 0
        torch.cuda.synchronize()
        same = bool(torch.equal(d_new, d_prev))
        reps = 20 if B >= 16384 else 100
        t = [event_time_ms(new, sp, f, reps) * 1e3 for f in (f_prev, f_new, f_prev, f_new)]
        print(f"read bwd {Hh}x{Ww}/{h}x{w} T={T} B={B:6d} bitwise_equal={same} | prev {t[0]:8.2f} {t[2]:8.2f} us | new {t[1]:8.2f} {t[3]:8.2f} us")
        del img, dgl, where
