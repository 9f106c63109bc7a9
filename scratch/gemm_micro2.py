import sys, ctypes, time, torch
sys.path.insert(0, '.')
from attend_infer_repeat_amd import _lib, hip as H
L = H.lib()
s = torch.cuda.Stream(); sp = ctypes.c_void_p(s.cuda_stream)
def graph_time(fn, nodes=48, reps=200):
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        _lib.check(L.air_graph_begin_capture(sp))
        for _ in range(nodes): fn()
        exe = ctypes.c_void_p(); _lib.check(L.air_graph_end_capture(sp, ctypes.byref(exe)))
        for _ in range(10): L.air_graph_launch(exe, sp)
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): L.air_graph_launch(exe, sp)
        s.synchronize(); dt = (time.perf_counter() - t0) / reps / nodes * 1e6
        L.air_graph_destroy(exe)
    return dt
keep = []
def run(descs):
    arr = (_lib.AirGemmDesc * len(descs))(*descs); keep.append(arr)
    return lambda: _lib.check(L.air_gemm_grouped(arr, len(descs), sp))
def D(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, epi=0, bias=None):
    return _lib.AirGemmDesc(ta, tb, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc, bias.data_ptr() if bias is not None else None, epi, None, 0, 0.0, None)
dg = torch.randn(64, 1024, device='cuda'); Wh = torch.randn(256, 1024, device='cuda'); out = torch.empty(64, 256, device='cuda')
tiny_a = torch.randn(16, 16, device='cuda'); tiny_b = torch.randn(16, 16, device='cuda'); tiny_c = torch.empty(16, 16, device='cuda')
print("NT 64x256x1024 alone (KW=16 path): %.2f us" % graph_time(run([D(0, 1, 64, 256, 1024, dg, 1024, Wh, 1024, out, 256)])))
print("NT 64x256x1024 + tiny 16x16x16 (forces KW=4): %.2f us" % graph_time(run([D(0, 1, 64, 256, 1024, dg, 1024, Wh, 1024, out, 256), D(0, 0, 16, 16, 16, tiny_a, 16, tiny_b, 16, tiny_c, 16)])))
x = torch.randn(64, 2500, device='cuda'); W = torch.randn(2500, 256, device='cuda'); b = torch.randn(256, device='cuda'); y = torch.empty(64, 256, device='cuda')
print("NN 64x256x2500 alone (KW=16): %.2f us" % graph_time(run([D(0, 0, 64, 256, 2500, x, 2500, W, 256, y, 256, 2, b)])))
print("NN 64x256x2500 + tiny (KW=4): %.2f us" % graph_time(run([D(0, 0, 64, 256, 2500, x, 2500, W, 256, y, 256, 2, b), D(0, 0, 16, 16, 16, tiny_a, 16, tiny_b, 16, tiny_c, 16)])))
ws = torch.empty(8 << 20, device='cuda')
f = lambda: _lib.check(L.air_gemm(0, 0, 64, 256, 2500, H._p(x), 2500, H._p(W), 256, H._p(y), 256, H._p(b), 2, None, 0, 0.0, None, H._p(ws), ctypes.c_size_t(ws.numel() * 4), sp))
print("NN 64x256x2500 air_gemm split-K (2 launches): %.2f us" % graph_time(f))
for K in (256, 512, 1024, 2048):
    A = torch.randn(64, K, device='cuda'); Bm = torch.randn(256, K, device='cuda')
    print(f"NT 64x256x{K}: %.2f us" % graph_time(run([D(0, 1, 64, 256, K, A, K, Bm, K, out, 256)])))
