"""bench.py's N > 1 control flow on a single GPU: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), sharing GPU 0 over the gloo backend
(AIR_BENCH_SHARE_GPU=1).  Not a measurement -- it checks that the script cannot deadlock or mis-report with more than one rank:
every rank runs every step that contains a collective, the clock is the maximum over ranks, rank 0 prints one JSON line with the
rank count the communication layer reports, and both replicas end with identical parameters (checked inside DataParallelEngine's
own tests; here: finite)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("launcher", ["driver", "self"])
def test_bench_two_ranks_sharing_one_gpu(gpu_device, launcher):
    """launcher = "driver": the command line the driver uses for N > 1; "self": a plain `python bench.py --gpus 2`, which must
    re-execute itself under torch.distributed.run with two ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, AIR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--no-sweep", "--no-cpu-baseline"]
    if launcher == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["collective"] == "torch-overlap" and d["config"]["dist_world_size"] == 2
    assert d["config"]["params_finite_after_run"] is True and d["config"]["replicas_in_sync_after_run"] is True
    assert d["value"] > 0 and "NOT A MEASUREMENT" in d["data"]


def test_bench_refuses_more_gpus_than_the_node_has(gpu_device):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AIR_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "exposes" in (out.stderr + out.stdout)
