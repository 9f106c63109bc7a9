"""per-replay overhead: time/step when one hipGraph holds 1, 2, 4 train steps"""
import sys, time, ctypes
import torch
sys.path.insert(0, ".")
from attend_infer_repeat_amd import _lib, hip as H
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from attend_infer_repeat_amd.data import synthetic_multi_mnist

eng = AIREngine(EngineConfig(), 64, seed=1, keep_canvas_steps=False)
imgs, _ = synthetic_multi_mnist(64, (50, 50), 2, seed=0)
eng.set_obs(torch.from_numpy(imgs).cuda())
L = H.lib(); sp = eng._sp()
for k in (1, 2, 4, 8):
    g = eng._capture_plans([eng._plan_fwd_train, eng._plan_bwd, eng._plan_opt] * k)
    for _ in range(50):
        L.air_graph_launch(g, sp)
    eng.stream.synchronize()
    n = 2000 // k
    t0 = time.perf_counter()
    for _ in range(n):
        L.air_graph_launch(g, sp)
    eng.stream.synchronize()
    dt = time.perf_counter() - t0
    print(f"{k} steps/graph: {dt / (n * k) * 1e6:.2f} us/step")
    L.air_graph_destroy(g)
