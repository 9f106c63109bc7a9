#!/usr/bin/env python
"""Developer probe: air_st_read_fwd of the library in the tree against a copy of the previous build
(tools/kbench/bin/libair_hip_prev.so) on the same buffers, alternating, us per launch (HIP events, back to back)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attend_infer_repeat_amd import hip as H
from bench import event_time_ms, HBM_PEAK_GBS

new = H.lib()
prev_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kbench", "bin", "libair_hip_prev.so")
prev = ctypes.CDLL(prev_path)
P, I = ctypes.c_void_p, ctypes.c_int
prev.air_st_read_fwd.argtypes = [P, P, P, I, I, I, I, I, I, P]; prev.air_st_read_fwd.restype = I
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(stream.cuda_stream)
p = H._p
cases = [((50, 50), (20, 20), 3, [64, 8192, 65536]), ((100, 100), (28, 28), 5, [64, 8192, 65536])]
if os.environ.get("READ_AB_FULL"):
    cases = [((50, 50), (20, 20), 3, [64, 1024, 8192, 65536]), ((50, 50), (20, 20), 1, [24576, 196608]),
             ((100, 100), (28, 28), 5, [64, 1024, 8192, 65536]), ((100, 100), (28, 28), 1, [24576])]
for (Hh, Ww), (h, w), T, batches in cases:
    for B in batches:
        n = T * B
        img = torch.rand(B, Hh, Ww, device=dev)
        where = torch.empty(n, 4, device=dev)
        where[:, 0] = 0.45 + 0.2 * torch.rand(n, device=dev); where[:, 2] = 0.45 + 0.2 * torch.rand(n, device=dev)
        where[:, 1] = 0.6 * torch.rand(n, device=dev) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev) - 0.3
        # (ONE output buffer for the timed launches: where a 300 MB buffer lands moves a streaming kernel by +-10 %)
        o_new = torch.empty(n, h, w, device=dev); o_prev = torch.empty(n, h, w, device=dev)
        f_new = lambda: new.air_st_read_fwd(p(img), p(where), p(o_new), n, B, Hh, Ww, h, w, sp)
        f_prev = lambda: prev.air_st_read_fwd(p(img), p(where), p(o_prev), n, B, Hh, Ww, h, w, sp)
        torch.cuda.synchronize()
        assert f_new() == 0 and f_prev() == 0
        torch.cuda.synchronize()
        same = bool(torch.equal(o_new, o_prev))
        reps = 20 if B >= 16384 else 100
        f_prev = lambda: prev.air_st_read_fwd(p(img), p(where), p(o_new), n, B, Hh, Ww, h, w, sp)
        t = [event_time_ms(new, sp, f, reps) * 1e3 for f in (f_prev, f_new, f_prev, f_new)]
        minimal = 4 * (B * Hh * Ww + n * (h * w + 4))
        fr = [minimal / (x * 1e-6) / 1e9 / HBM_PEAK_GBS for x in t]
        print(f"{Hh}x{Ww}/{h}x{w} T={T} B={B:6d} bitwise_equal={same} | prev {t[0]:8.2f} {t[2]:8.2f} us ({fr[0]:.3f} {fr[2]:.3f}) | new {t[1]:8.2f} {t[3]:8.2f} us ({fr[1]:.3f} {fr[3]:.3f})")
        del img, where, o_new, o_prev
