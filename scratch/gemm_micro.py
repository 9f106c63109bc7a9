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
def D(ta, tb, M, N, K, A, B, C, bias=None, epi=0, aux=None):
    lda = A.shape[1]; ldb = B.shape[1]
    return _lib.AirGemmDesc(ta, tb, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), N, bias.data_ptr() if bias is not None else None, epi, aux.data_ptr() if aux is not None else None, N if aux is not None else 0, 0.0, None)
print("NN M=192 N=256 vs K (bias+elu epilogue), ping-pong buffers so each node depends on the previous")
for K in (16, 64, 256, 512, 1024):
    X = [torch.randn(192, max(K, 256), device='cuda') for _ in range(2)]
    W = torch.randn(K, 256, device='cuda'); b = torch.randn(256, device='cuda')
    state = {"i": 0}
    def fn():
        i = state["i"]; state["i"] ^= 1
        A = X[i][:, :K]
        d = _lib.AirGemmDesc(0, 0, 192, 256, K, A.data_ptr(), X[i].shape[1], W.data_ptr(), 256, X[i ^ 1].data_ptr(), X[i ^ 1].shape[1], b.data_ptr(), 2, None, 0, 0.0, None)
        arr = (_lib.AirGemmDesc * 1)(d); keep.append(arr)
        _lib.check(L.air_gemm_grouped(arr, 1, sp))
    keep = []
    print(f"  K={K:5d}: {graph_time(fn):6.2f} us/node")
print("tile-count scaling: NN Mx256x256")
for M in (16, 64, 192, 768, 3072):
    X = [torch.randn(M, 256, device='cuda') for _ in range(2)]
    W = torch.randn(256, 256, device='cuda'); b = torch.randn(256, device='cuda')
    state = {"i": 0}; keep = []
    def fn():
        i = state["i"]; state["i"] ^= 1
        d = _lib.AirGemmDesc(0, 0, M, 256, 256, X[i].data_ptr(), 256, W.data_ptr(), 256, X[i ^ 1].data_ptr(), 256, b.data_ptr(), 2, None, 0, 0.0, None)
        arr = (_lib.AirGemmDesc * 1)(d); keep.append(arr)
        _lib.check(L.air_gemm_grouped(arr, 1, sp))
    print(f"  M={M:5d}: {graph_time(fn):6.2f} us/node")
print("reference: trivial kernel (air_fill 256 floats)")
t = torch.empty(256, device='cuda')
print(f"  fill: {graph_time(lambda: L.air_fill(H._p(t), ctypes.c_size_t(256), 1.0, sp)):6.2f} us/node")
x = torch.randn(192, 256, device='cuda'); y = torch.empty_like(x)
print(f"  axpby 49k: {graph_time(lambda: L.air_axpby(H._p(x), 2.0, None, 0.0, H._p(y), ctypes.c_size_t(x.numel()), sp)):6.2f} us/node")
