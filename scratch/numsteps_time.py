"""isolated cost of the float64 presence / num-steps kernels (role B of the attend launches)"""
import sys, ctypes
import torch
sys.path.insert(0, ".")
from attend_infer_repeat_amd import hip as H
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from bench import event_time_ms
eng = AIREngine(EngineConfig(), 64, seed=1)
eng.forward(); eng.backward(); torch.cuda.synchronize()
L = H.lib(); sp = eng._sp(); p = H._p
T, B = eng.T, eng.B
f = lambda: L.air_presence_numsteps_fwd(p(eng.st.out[-1]), p(eng.u_pres), 0.75, 1e-3, p(eng.prior_dev), p(eng.presence_prob),
                                        p(eng.presence), p(eng.q_n), p(eng.kl_n), p(eng.logp), p(eng.step_w), T, B, sp)
b = lambda: L.air_numsteps_presence_bwd(p(eng.presence_prob), p(eng.presence), p(eng.prior_dev), 1.0 / B, p(eng.kl_what_row),
                                        p(eng.kl_where_row), 1.0 / B, p(eng.dlogp), p(eng.st.out[-1]), 0.75, 1e-3,
                                        p(eng.st.g[-1]), T, B, sp)
e = lambda: L.air_fill(p(eng.dc_a), ctypes.c_size_t(64), 0.0, sp)
for name, fn in (("empty-ish fill", e), ("presence_numsteps_fwd", f), ("numsteps_presence_bwd", b)):
    print(name, round(event_time_ms(L, sp, fn, 500) * 1e3, 2), "us")
