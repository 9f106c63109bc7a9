"""Behavioural check of the train step over a real stretch of training ("learning curves match statistically").

The per-step parity tests compare ONE update with identical noise.  Here the engine and the CPU oracle each train for 1500
updates from the same initial weights on the same stream of batches (procedural digit templates through the reference's own
dataset generator, data.create_multi_mnist), each with its OWN noise, through the hold-out and the start of the num-steps prior
anneal (model.py:106-124: the prior starts moving at step 1000).  A sign or weight error in any term of the objective or in the
optimiser moves these curves apart systematically; noise moves them by the spread the engine shows against itself under a second
noise seed.  Asserted: from the checkpoint at 1000 updates on, the oracle's smoothed reconstruction term, KL terms and mean step
count lie within a band around the engine's, the band being max(absolute floor, 4 x the engine's own seed-to-seed difference);
before that -- while the reconstruction term falls by hundreds of nats within a few hundred updates, at a moment the noise
decides (measured: -367 / -319 / -119 at update 500 for the two engine seeds and the oracle, -538 / -445 / -530 at 1500) -- only
that all three are on their way down.  (Round-3 run: profiles/r03_dynamics_report.json.)"""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import air_oracle as O

pytestmark = pytest.mark.gpu

N_STEPS, EVERY, B = 1500, 250, 64
KEYS = ("rec_loss", "kl_num_steps", "kl_what", "kl_where", "num_step")


def _batches():
    from attend_infer_repeat_amd.data import procedural_multi_mnist
    d = procedural_multi_mnist(4096, seed=5, n_templates=1000)
    imgs = torch.from_numpy(d["imgs"].astype(np.float32) / 255.0)
    idx = torch.randint(0, imgs.shape[0], (N_STEPS, B), generator=torch.Generator().manual_seed(6))
    return imgs, idx


def _smooth(rows):
    """mean of each logged quantity over the window that ends at a checkpoint"""
    out = []
    for c in range(EVERY, N_STEPS + 1, EVERY):
        win = rows[c - EVERY:c]
        out.append({k: float(np.mean([r[k] for r in win])) for k in KEYS})
    return out


def _train_engine(params, imgs, idx, seed):
    from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
    ocfg = O.AIRConfig()
    fields = {f.name for f in dataclasses.fields(EngineConfig)}
    eng = AIREngine(EngineConfig(**{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields}), B, seed=seed)
    eng.load_parameters(params)
    dev = imgs.cuda()
    eng.set_obs(dev[idx[0]].reshape(B, -1))
    eng.capture()
    rows = []
    for s in range(N_STEPS):
        eng.train_step(dev[idx[s]].reshape(B, -1))
        o = eng.outputs()
        rows.append({"rec_loss": o["rec_loss"].item(), "kl_num_steps": o["kl_num_steps"].item(), "kl_what": o["kl_what"].item(),
                     "kl_where": o["kl_where"].item(), "num_step": o["num_step_per_sample"].mean().item()})
    assert torch.isfinite(eng.flat_params).all()
    return _smooth(rows)


def _train_oracle(params, imgs, idx, seed):
    ocfg = O.AIRConfig()
    p = {k: v.clone() for k, v in params.items()}
    slots = O.rmsprop_init(p)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    rows = []
    for s in range(N_STEPS):
        noise = O.make_noise(ocfg, B, seed=seed * 100003 + s)
        res, _ = O.train_step(p, slots, ocfg, imgs[idx[s]], noise, global_step=s)
        rows.append({"rec_loss": res["rec_loss"].item(), "kl_num_steps": res["kl_num_steps"].item(),
                     "kl_what": res["kl_what"].item(), "kl_where": res["kl_where"].item(),
                     "num_step": res["presence"].sum(0).mean().item()})
    return _smooth(rows)


def test_engine_and_oracle_learning_curves_agree(gpu_device):
    import json
    import os
    imgs, idx = _batches()
    params = O.init_params(O.AIRConfig(), seed=3)
    e1 = _train_engine(params, imgs, idx, seed=11)
    e2 = _train_engine(params, imgs, idx, seed=12)
    orc = _train_oracle(params, imgs, idx, seed=13)
    floors = {"rec_loss": 12.0, "kl_num_steps": 0.6, "kl_what": 1.5, "kl_where": 0.6, "num_step": 0.35}
    report = {"checkpoints": list(range(EVERY, N_STEPS + 1, EVERY)), "engine_seed_a": e1, "engine_seed_b": e2, "oracle": orc}
    path = os.environ.get("AIR_DYNAMICS_REPORT")
    if path:
        with open(path, "w") as f:
            json.dump(report, f, indent=1)
    # training moved: the reconstruction term improved by hundreds of nats from the first window to the last
    assert e1[-1]["rec_loss"] < e1[0]["rec_loss"] - 50 and orc[-1]["rec_loss"] < orc[0]["rec_loss"] - 50
    for c, (a, b, o) in enumerate(zip(e1, e2, orc)):
        if report["checkpoints"][c] < 1000:                  # the transient: direction only
            if c > 0:
                assert o["rec_loss"] < orc[c - 1]["rec_loss"] and a["rec_loss"] < e1[c - 1]["rec_loss"]
            continue
        for k in KEYS:
            # The oracle (a THIRD noise seed) must lie inside the band the engine's own two seeds span, widened by 3x the metric's
            # floor on either side.  (Rounds 3-4 compared it with the MIDPOINT of the two seeds under a band of 4x their largest late
            # spread -- ADVICE r04: up to 4.8 in kl_where, a real drift could hide in it; a midpoint is also meaningless when the two
            # REINFORCE trajectories part ways: round 5 saw rec_loss -528 / -380 at update 1000 with the oracle at -532.)
            lo, hi = min(a[k], b[k]), max(a[k], b[k])
            dist = max(0.0, lo - o[k], o[k] - hi)
            assert dist <= 3.0 * floors[k], (report["checkpoints"][c], k, o[k], a[k], b[k], 3.0 * floors[k])
    # the annealed num-steps prior enters both identically: once it moves, the KL of the step count follows it to the digit
    assert abs(orc[-1]["kl_num_steps"] - e1[-1]["kl_num_steps"]) < 0.15
