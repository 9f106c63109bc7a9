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

from attend_infer_repeat_amd.distributed import free_rendezvous_port as D_free_port   # below the ephemeral range

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("launcher", ["driver", "self"])
def test_bench_two_ranks_sharing_one_gpu(gpu_device, launcher):
    """launcher = "driver": the command line the driver uses for N > 1; "self": a plain `python bench.py --gpus 2`, which must
    re-execute itself under torch.distributed.run with two ranks."""
    port = D_free_port()
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
    c = d["config"]
    assert c["dist_world_size"] == 2
    # one invocation times the safe protocol, the bare collective and the overlapped protocol; the headline is the fastest VALIDATED one
    ab = c["protocol_ab"]
    # (round 5: with the ranks sharing a GPU the hipIpc protocol -- no library collective -- is timed too; on real GPUs it is opt-in)
    assert set(ab) == {"torch-split", "torch-overlap", "ipc-rsag"}, ab
    for name, r in ab.items():
        assert r["replicas_in_sync_after_run"] is True and r["params_finite_after_run"] is True and r["images_per_sec"] > 0, (name, r)
    assert c["collective"] in ab and d["value"] == max(r["images_per_sec"] for r in ab.values())
    assert c["allreduce_us"] > 0 and 4 * 2629118 <= c["allreduce_bytes"] <= 4 * 2640000 and c["allreduce_busbw_GBs"] > 0
    assert d["roofline_comm"]["bound"] == "xgmi" and d["roofline"] is not None
    assert c["params_finite_after_run"] is True and c["replicas_in_sync_after_run"] is True
    assert d["value"] > 0 and "NOT A MEASUREMENT" in d["data"]


def test_bench_watchdog_prints_the_safe_protocols_line_when_a_later_protocol_hangs(gpu_device):
    """A protocol that never returns (AIR_BENCH_FAKE_HANG: the overlapped one sleeps forever on every rank) must not cost the run its
    result: every rank's watchdog ends the process with exit code 0 and rank 0 prints the line of the torch-split measurement that
    had already finished -- what `bench.py --gpus N` does if the overlapped protocol deadlocks on its first contact with RCCL."""
    port = D_free_port()
    env = dict(os.environ, AIR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", AIR_BENCH_FAKE_HANG="torch-overlap",
               AIR_BENCH_PROTOCOL_TIMEOUT_S="25")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--no-sweep",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["collective"] == "torch-split" and set(d["config"]["protocol_ab"]) == {"torch-split"}
    assert "did not finish" in d["config"]["protocol_note"] and d["value"] > 0 and d["n_gpus"] == 2
    assert d["config"]["allreduce_us"] > 0                      # (the bare collective ran before the hanging protocol)


def test_bench_refuses_more_gpus_than_the_node_has(gpu_device):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AIR_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "exposes" in (out.stderr + out.stdout)
