#!/usr/bin/env python
"""what does a GEMM node of the replayed train step pay for finding its WEIGHTS (and its code) cold?

In the step graph every dense node runs 5-6 us against ~3 us when the same launch is repeated back to back.  A node's activation
operand is necessarily cold (the launch in front of it wrote it; it arrives through the Infinity Cache), so the part a prefetch
could remove is what the node pays for its weight tiles -- last touched a step ago, by then evicted from the 4 MB L2 of the XCD the
tile runs on -- and for its instructions.  This probe measures that part IN SITU, as an upper bound for any prefetch scheme: the
graph is re-captured with every weight-reading GEMM launch preceded by a twin -- the same kernel on the same grid (so the same
tile -> XCD placement) reading the same weights, but with dummy activation / output buffers -- i.e. the perfect prefetcher: right
before the real node runs, every weight byte and every instruction it needs is resident exactly where it will look.

  mode base : the step as shipped                                   (run under rocprofv3 --kernel-trace)
  mode warm : every weight-reading air_gemm_grouped / air_gemm launch doubled as described

tools/rocpd_summary.py --by-position on both traces gives the per-node durations; this script prints which positions of the warm
graph are twins (so that the real nodes can be read off) to stdout as JSON.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIR_OPT_FOLD", "0")

import torch  # noqa: E402

from attend_infer_repeat_amd import _lib, hip as H  # noqa: E402
from attend_infer_repeat_amd.data import synthetic_multi_mnist  # noqa: E402
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "base"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    dev = torch.device("cuda", 0)
    B = 64
    eng = AIREngine(EngineConfig(), B, device=dev, seed=1)
    imgs, _ = synthetic_multi_mnist(B, (50, 50), max_objects=2, seed=0)
    eng.set_obs(torch.from_numpy(imgs).to(dev))
    plans = eng._single_gpu_step_plans()
    p0, p1 = eng.flat_params.data_ptr(), eng.flat_params.data_ptr() + 4 * eng.n_total
    in_params = lambda x: x is not None and p0 <= int(x) < p1
    keep, twins, flat = [], [], []
    dummy_cache = {}

    def dummy(nbytes, tag):
        key = (tag, nbytes)
        if key not in dummy_cache:
            dummy_cache[key] = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        return dummy_cache[key].data_ptr()

    for pl in plans:
        for e in pl:
            if mode == "warm" and e[2] == "air_gemm_grouped":
                arr, n = e[1]
                warm = []
                for i in range(n):
                    d = arr[i]
                    w = _lib.AirGemmDesc()
                    for f, _t in _lib.AirGemmDesc._fields_:
                        setattr(w, f, getattr(d, f))
                    rows_a = (d.K if d.ta else d.M)
                    rows_b = (d.N if d.tb else d.K)
                    if not in_params(d.A):
                        w.A = dummy(4 * rows_a * d.lda, ("A", len(flat), i))
                    if not in_params(d.B):
                        w.B = dummy(4 * rows_b * d.ldb, ("B", len(flat), i))
                    w.C = dummy(4 * d.M * d.ldc, ("C", len(flat), i))
                    if d.colsum:
                        w.colsum = dummy(4 * d.N, ("col", len(flat), i))
                    if d.aux and not in_params(d.aux):
                        w.aux = dummy(4 * d.M * max(d.ldaux, 1), ("aux", len(flat), i))
                    if d.A2:
                        w.A2 = dummy(4 * rows_a * d.lda, ("A2", len(flat), i))
                    if d.a_out:
                        w.a_out = dummy(4 * rows_a * d.lda, ("ao", len(flat), i))
                    w.beta = 0.0
                    warm.append(w)
                # the twin runs the SAME kernel on the same grid; problems that read no weights (weight gradients riding in the launch)
                # get dummy operands too, so that nothing but weights and code is warmed
                if any(in_params(arr[i].A) or in_params(arr[i].B) for i in range(n)):
                    warr = (_lib.AirGemmDesc * n)(*warm)
                    keep.append(warr)
                    twins.append(len(flat))
                    flat.append((e[0], (warr, n), "twin:" + e[2]))
            flat.append(e)
    L = H.lib()
    sp = eng._sp()
    _lib.check(L.air_graph_begin_capture(sp))
    for e in flat:
        st = e[0](*e[1], sp)
        assert st == 0, (e[2], st)
    exe = ctypes.c_void_p()
    _lib.check(L.air_graph_end_capture(sp, ctypes.byref(exe)))
    for _ in range(50):
        _lib.check(L.air_graph_launch(exe, sp))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        _lib.check(L.air_graph_launch(exe, sp))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"mode": mode, "nodes": len(flat), "twin_positions": twins, "names": [e[2] for e in flat],
                      "ms_per_replay": round(ms, 4), "finite": bool(torch.isfinite(eng.flat_params).all().item())}))


if __name__ == "__main__":
    main()
