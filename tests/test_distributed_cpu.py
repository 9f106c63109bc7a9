"""World-size-2 gloo tests (CPU) of the data-parallel host logic: batch sharding, per-rank seeds, the single flat
gradient all-reduce and the 1/world scaling.  The gradient producer here is the CPU oracle (test infrastructure): the
test checks the DP *semantics* of SURVEY 8e -- the averaged gradient equals the mean of the per-shard reference
gradients -- with the exact collective code the GPU path uses."""
import os
import socket
import sys

import numpy as np
import pytest

from attend_infer_repeat_amd.distributed import free_rendezvous_port as D_free_port   # below the ephemeral range
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    return D_free_port()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from attend_infer_repeat_amd import distributed as D
    from oracle import air_oracle as O
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3)
    global_batch = 8
    lo, hi = D.shard_batch(global_batch, rank, world)
    obs, _ = O.synthetic_batch(cfg, global_batch, seed=0)
    params = O.init_params(cfg, seed=1 + rank)                         # deliberately different before the broadcast
    names = list(params)
    flat = torch.cat([params[k].reshape(-1) for k in names])
    D.broadcast_parameters(flat, 0)
    off = 0
    for k in names:
        n = params[k].numel(); params[k] = flat[off:off + n].reshape(params[k].shape).clone(); off += n
    noise = O.make_noise(cfg, hi - lo, seed=D.rank_seed(5, rank) % (2 ** 31))
    _, grads = O.forward_backward(params, cfg, obs[lo:hi], noise)
    flat_g = torch.cat([grads[k].reshape(-1) for k in names])
    local = flat_g.clone()
    D.allreduce_gradients(flat_g, average=False)                       # the single collective of the step
    torch.save(dict(local=local, summed=flat_g, flat_params=flat, lo=lo, hi=hi, seed=D.rank_seed(5, rank)),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 4, 4, 8)
    assert res[0]["seed"] != res[1]["seed"]
    assert torch.equal(res[0]["flat_params"], res[1]["flat_params"])   # replicas start identical
    expected = res[0]["local"] + res[1]["local"]
    for r in res:
        assert torch.allclose(r["summed"], expected, rtol=1e-6, atol=1e-7)
    assert not torch.equal(res[0]["local"], res[1]["local"])
    # mean of the per-rank gradients == what RMSProp sees with grad_scale = 1/world
    mean = expected / world
    assert torch.isfinite(mean).all() and mean.abs().max() > 0


class _StandInEngine(object):
    """What DataParallelEngine needs from an engine, on CPU tensors: a flat parameter / gradient buffer, capture() and
    train_step(obs, allreduce).  The "gradient" of step s on rank r is a known vector, so the update every replica must end
    up with is known in closed form."""

    def __init__(self, n, rank):
        self.device = torch.device("cpu")
        self.world_size = 1
        self.flat_params = torch.full((n,), float(rank + 1))          # deliberately different before the broadcast
        self.flat_grads = torch.zeros(n)
        self.rank, self.step, self.captured, self.reduced = rank, 0, [], []

    def capture(self, **kw):
        self.captured.append(kw)

    def synchronize(self):
        pass

    def train_step(self, obs=None, allreduce=None):
        self.flat_grads.copy_(torch.arange(self.flat_grads.numel(), dtype=torch.float32) * (self.rank + 1) + self.step)
        if allreduce is not None:
            self.reduced.append(self.flat_grads.numel())
            allreduce(self.flat_grads)
        self.flat_params.sub_(0.1 * self.flat_grads / self.world_size)  # grad_scale = 1/world, like air_step_epilogue
        self.step += 1


def _dp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from attend_infer_repeat_amd import distributed as D
    D.init_from_env(backend="gloo")
    eng = _StandInEngine(7, rank)
    # rank 1 ASKS for the captured-RCCL protocol through the environment, rank 0 does not: an engine that is not on a GPU must end
    # up in the host-issued protocol on every rank alike (the own-communicator protocols are GPU-only and opt-in)
    if rank == 1:
        os.environ["AIR_DP_COLLECTIVE"] = "rccl-captured"
    dp = D.DataParallelEngine(eng)
    assert dp.rccl_nranks is None and dp.comm is None
    start = eng.flat_params.clone()
    for _ in range(3):
        dp.train_step()
    in_sync = dp.replicas_in_sync()                         # collective: identical bits on every rank after three updates
    if rank == 1:
        eng.flat_params[3] += 1e-6                          # one replica drifts by less than an ulp-level tolerance would catch
    drift_seen = not dp.replicas_in_sync()
    torch.save(dict(start=start, params=eng.flat_params, captured=eng.captured, reduced=eng.reduced,
                    collective=dp.collective, world=eng.world_size, in_sync=in_sync, drift_seen=drift_seen),
               os.path.join(out_dir, f"dp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_engine_control_flow_two_ranks(tmp_path):
    """DataParallelEngine itself over gloo, world 2: parameters are broadcast from rank 0, the engine is told the world size
    (-> grad_scale), a CPU engine gets the two-graph protocol (capture(split_optimizer=True)) with exactly ONE all-reduce of
    the whole flat bucket per step, and both replicas apply the mean gradient."""
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"dp{r}.pt")) for r in range(world)]
    for r in res:
        assert r["collective"] == "torch-split" and r["world"] == 2
        assert r["captured"] == [dict(split_optimizer=True)]
        assert r["reduced"] == [7, 7, 7]
        assert torch.equal(r["start"], torch.full((7,), 1.0))           # rank 0's parameters everywhere
        assert r["in_sync"] is True and r["drift_seen"] is True         # replicas_in_sync(): equal bits yes, a drifted replica no
    expect = torch.full((7,), 1.0)
    for s in range(3):
        g = sum(torch.arange(7, dtype=torch.float32) * (rk + 1) + s for rk in range(world))
        expect = expect - 0.1 * g / world
    assert torch.allclose(res[0]["params"], expect, rtol=1e-6)
    assert torch.allclose(res[0]["params"], res[1]["params"], rtol=1e-5)   # (rank 1 perturbed one element after the in-sync check)


def test_collective_name_is_validated():
    from attend_infer_repeat_amd import distributed as D
    eng = _StandInEngine(3, 0)
    with pytest.raises(ValueError):
        D.DataParallelEngine(eng, collective="allreduce-in-a-thread")
    dp = D.DataParallelEngine(eng, collective="captured")              # alias; world 1: nothing to reduce
    assert dp.collective == "none" and dp.world == 1 and eng.captured == [{}]


def test_shard_and_seed_helpers():
    from attend_infer_repeat_amd import distributed as D
    assert [D.shard_batch(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        D.shard_batch(10, 0, 4)
    seeds = {D.rank_seed(3, r) for r in range(8)}
    assert len(seeds) == 8 and all(s >= 0 for s in seeds)
    g = torch.ones(5)
    assert D.allreduce_gradients(g) is g                               # world 1 / uninitialised: no-op
