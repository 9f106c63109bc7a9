import os, sys, time, socket, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from attend_infer_repeat_amd import distributed as D
from attend_infer_repeat_amd.data import synthetic_multi_mnist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
imgs, _ = synthetic_multi_mnist(64, (50, 50), 2, 0)
def run(mode):
    eng = AIREngine(EngineConfig(), 64, keep_canvas_steps=False); eng.set_obs(torch.from_numpy(imgs).cuda())
    if mode == "single": eng.capture(); ar = None
    elif mode == "split": eng.capture(split_optimizer=True); ar = lambda g: (dist.all_reduce(g), None)[1]
    else: eng.capture(split_optimizer=True, bucketed=True); ar = lambda g: dist.all_reduce(g, async_op=True)
    for _ in range(30): eng.train_step(allreduce=ar)
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(300): eng.train_step(allreduce=ar)
    eng.synchronize(); return (time.perf_counter() - t0) / 300 * 1e6
for m in ("single", "split", "bucketed"):
    print(m, "%.1f us/step" % run(m))
dist.destroy_process_group()
