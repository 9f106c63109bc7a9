import sys, ctypes, torch
sys.path.insert(0, '.')
from attend_infer_repeat_amd import hip as H
lib = H.lib()
dev = torch.device('cuda', 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "one"
Hh = Ww = 50; h = w = 20
if mode == "one":
    n_img, n = 196608, 196608
else:
    n_img, n = 65536, 196608
img = torch.rand(n_img, Hh, Ww, device=dev)
where = torch.empty(n, 4, device=dev)
where[:, 0] = 0.45 + 0.2 * torch.rand(n, device=dev); where[:, 2] = 0.45 + 0.2 * torch.rand(n, device=dev)
where[:, 1] = 0.6 * torch.rand(n, device=dev) - 0.3; where[:, 3] = 0.6 * torch.rand(n, device=dev) - 0.3
out = torch.empty(n, h, w, device=dev)
torch.cuda.synchronize()
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    lib.air_st_read_fwd(H._p(img), H._p(where), H._p(out), n, n_img, Hh, Ww, h, w, sp)
torch.cuda.synchronize()
print("algorithmic bytes per launch", 4 * (Hh * Ww + h * w + 4) * n, "actual minimum traffic", 4 * (Hh * Ww * n_img + (h * w + 4) * n))
