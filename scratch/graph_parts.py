import sys, ctypes, time, torch
sys.path.insert(0, '.')
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from attend_infer_repeat_amd import _lib, hip as H
from attend_infer_repeat_amd.data import synthetic_multi_mnist
L = H.lib()
eng = AIREngine(EngineConfig(), 64, keep_canvas_steps=False)
imgs, _ = synthetic_multi_mnist(64, (50, 50), 2, 0)
eng.set_obs(torch.from_numpy(imgs).cuda())
for _ in range(3): eng.train_step()
eng.synchronize()
sp = eng._sp()
def cap(plans):
    _lib.check(L.air_graph_begin_capture(sp))
    for pl in plans: eng._run(pl, sp)
    exe = ctypes.c_void_p(); _lib.check(L.air_graph_end_capture(sp, ctypes.byref(exe))); return exe
def timeit(exe, n=300):
    for _ in range(20): L.air_graph_launch(exe, sp)
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(n): L.air_graph_launch(exe, sp)
    eng.synchronize(); return (time.perf_counter() - t0) / n * 1e6
parts = {"fwd": [eng._plan_fwd_noise], "bwd": [eng._plan_bwd], "opt": [eng._plan_opt], "all": [eng._plan_fwd_noise, eng._plan_bwd, eng._plan_opt]}
for k, pl in parts.items():
    nl = sum(len(p) for p in pl)
    us = timeit(cap(pl))
    print(f"{k:4s} launches {nl:3d}  {us:7.1f} us  ({us / nl:.2f} us/launch)")
# per-kind subsets inside fwd: only the grouped gemms / only the rest
fw = eng._plan_fwd_noise
g_only = [e for e in fw if e[2].startswith("air_gemm")]
rest = [e for e in fw if not e[2].startswith("air_gemm")]
for k, pl in (("fwd gemm only", g_only), ("fwd non-gemm", rest)):
    us = timeit(cap([pl])); print(f"{k:14s} launches {len(pl):3d} {us:7.1f} us ({us / len(pl):.2f} us/launch)")
bw = eng._plan_bwd
g_only = [e for e in bw if e[2].startswith("air_gemm")]
rest = [e for e in bw if not e[2].startswith("air_gemm")]
for k, pl in (("bwd gemm only", g_only), ("bwd non-gemm", rest)):
    us = timeit(cap([pl])); print(f"{k:14s} launches {len(pl):3d} {us:7.1f} us ({us / len(pl):.2f} us/launch)")
