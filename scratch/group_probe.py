"""time individual problems of the slow grouped launches (which member makes the group slow?)"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from attend_infer_repeat_amd import _lib, hip as H
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from bench import event_time_ms

eng = AIREngine(EngineConfig(), 64, seed=1)
eng.forward(); eng.backward(); torch.cuda.synchronize()
L = H.lib(); sp = eng._sp()
def t(fn, a):
    return event_time_ms(L, sp, lambda: fn(*a, sp), 200) * 1e3
for pname, plan in (("fwd", eng._plan_fwd_noise), ("bwd", eng._plan_bwd)):
    for i, (fn, a, name) in enumerate(plan):
        if name != "air_gemm_grouped":
            continue
        descs = list(a[0])
        whole = t(fn, a)
        parts = []
        for d in descs:
            arr = (_lib.AirGemmDesc * 1)(d)
            parts.append(t(L.air_gemm_grouped, (arr, 1)))
        sh = " | ".join("%s%s %dx%dx%d e%d%s%s" % ("T" if d.ta else "N", "T" if d.tb else "N", d.M, d.N, d.K, d.epilogue,
                                                    "+cs" if d.colsum else "", "+b%g" % d.beta if d.beta else "") for d in descs)
        print("%s %2d  %6.2f us  parts %s   %s" % (pname, i, whole, " ".join("%.2f" % x for x in parts), sh))
