#!/usr/bin/env python
"""Round 6: stand-alone time of air_mlp_dx_chain_bf16 on the configs[4] decoder chain (3072 rows, 400 -> 256 -> 256 -> 50) against the same three
products as per-layer grouped launches (bf16 data path), HIP events, back to back."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from attend_infer_repeat_amd import hip as H, _lib
lib = H.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(stream.cuda_stream)
rows, widths = 3072, (400, 256, 256, 50)
g_in = torch.randn(rows, widths[0], device=dev)
keep, layers = [], []
for l in range(3):
    n_in, n_out = widths[l], widths[l + 1]
    w = torch.randn(n_out, n_in, device=dev) / n_in ** 0.5
    w16 = w.to(torch.bfloat16).contiguous()
    aux = torch.randn(rows, n_out, device=dev) if l < 2 else None
    out = torch.empty(rows, n_out, device=dev); out16 = torch.empty(rows, n_out, dtype=torch.bfloat16, device=dev)
    layers.append((w, w16, aux, out, out16, n_in, n_out))
arr = (_lib.AirDxChain * 1)()
arr[0].g_in, arr[0].ld_in, arr[0].rows, arr[0].n_layers = g_in.data_ptr(), widths[0], rows, 3
for li, (w, w16, aux, out, out16, n_in, n_out) in enumerate(layers):
    y = arr[0].layer[li]
    y.w_bf16, y.aux, y.out, y.out_bf16 = w16.data_ptr(), (aux.data_ptr() if aux is not None else None), out.data_ptr(), out16.data_ptr()
    y.n_in, y.n_out, y.ldaux, y.ldout = n_in, n_out, n_out, n_out
us = bench.event_time_ms(lib, sp, lambda: lib.air_mlp_dx_chain_bf16(arr, 1, sp), 200) * 1e3
print("chain rows/slab", os.environ.get("AIR_DX_CHAIN_ROWS", "auto"), "us per launch %.2f" % us)
# the per-layer form
descs = []
src, src16 = g_in, None
for li, (w, w16, aux, out, out16, n_in, n_out) in enumerate(layers):
    d = _lib.AirGemmDesc()
    d.ta, d.tb, d.M, d.N, d.K = 0, 1, rows, n_out, n_in
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = src.data_ptr(), n_in, w.data_ptr(), n_in, out.data_ptr(), n_out
    d.epilogue, d.aux, d.ldaux = (H.EPI_MUL_DELU, aux.data_ptr(), n_out) if aux is not None else (H.EPI_NONE, None, 0)
    d.precision = 1
    d.A16, d.B16, d.C16 = (src16.data_ptr() if src16 is not None else None), w16.data_ptr(), out16.data_ptr()
    descs.append(d); src, src16 = out, out16
arrs = [(_lib.AirGemmDesc * 1)(d) for d in descs]
def per_layer():
    for a in arrs:
        lib.air_gemm_grouped(a, 1, sp)
us2 = bench.event_time_ms(lib, sp, per_layer, 200) * 1e3
print("three per-layer grouped launches: us %.2f" % us2)
