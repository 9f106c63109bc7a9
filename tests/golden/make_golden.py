#!/usr/bin/env python
"""Generates the golden fixtures in tests/golden/ from the CPU oracle (oracle/air_oracle.py, oracle/st_loops.c).

Run from the repo root in the build container:  python tests/golden/make_golden.py
The reference itself (Python 2 + TF 1.1 + Sonnet 1.1) cannot be imported here, so these vectors come from the
restatement; the only reference-held known answers (test/prior_test.py) are stored verbatim in prior_known_answers.npz.
Fixtures are data only: inputs, weights, noise, expected outputs / losses / gradients.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import air_oracle as O          # noqa: E402
from oracle import st_loops as C            # noqa: E402


def st_cases():
    rng = np.random.default_rng(1234)
    out = {}
    for name, (H, W, h, w, B) in {"c2": (50, 50, 20, 20, 4), "rect": (7, 5, 3, 4, 5), "tiny": (3, 3, 2, 2, 6)}.items():
        img = rng.random((B, H, W)).astype(np.float32)
        glm = rng.standard_normal((B, h, w)).astype(np.float32)
        where = np.stack([rng.uniform(0.2, 1.4, B) * rng.choice([-1, 1], B), rng.uniform(-.8, .8, B),
                          rng.uniform(0.2, 1.4, B), rng.uniform(-.8, .8, B)], 1).astype(np.float32)
        dread = rng.standard_normal((B, h, w)).astype(np.float32)
        dwrite = rng.standard_normal((B, H, W)).astype(np.float32)
        dwh_r, dimg = C.st_read_bwd(img.astype(np.float64), where.astype(np.float64), dread.astype(np.float64))
        dglm, dwh_w = C.st_write_bwd(glm.astype(np.float64), where.astype(np.float64), dwrite.astype(np.float64))
        out.update({f"{name}/img": img, f"{name}/glm": glm, f"{name}/where": where, f"{name}/dread": dread,
                    f"{name}/dwrite": dwrite, f"{name}/read": C.st_read_fwd(img, where, (h, w)),
                    f"{name}/write": C.st_write_fwd(glm, where, (H, W)), f"{name}/dwhere_read": dwh_r,
                    f"{name}/dimg": dimg, f"{name}/dglm": dglm, f"{name}/dwhere_write": dwh_w})
    np.savez_compressed(os.path.join(HERE, "st_cases.npz"), **out)


def model_case(name, cfg, B, gstep):
    params = O.init_params(cfg, seed=11, bias_std=0.15)
    obs, _ = O.synthetic_batch(cfg, B, seed=12)
    if max(cfg.img_size) < 10:
        obs = torch.tensor(np.random.default_rng(12).random((B,) + tuple(cfg.img_size)), dtype=torch.float32)
    noise = O.make_noise(cfg, B, seed=13)
    p64 = {k: v.double() for k, v in params.items()}
    res, grads = O.forward_backward(p64, cfg, obs.double(), {k: v.double() for k, v in noise.items()}, global_step=gstep)
    out = {"obs": obs.numpy(), "global_step": np.int64(gstep)}
    out.update({f"param/{k}": v.numpy() for k, v in params.items()})
    out.update({f"noise/{k}": v.numpy() for k, v in noise.items()})
    keep = ["canvas", "glimpse", "what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob",
            "presence", "final_canvas", "rec_loss_per_sample", "kl_num_steps_per_sample", "kl_what_per_sample",
            "kl_where_per_sample", "num_steps_posterior", "prior_step_weight", "num_steps_log_prob", "baseline",
            "rec_loss", "kl_num_steps", "kl_what", "kl_where", "loss", "reinforce_loss", "baseline_loss", "opt_loss"]
    out.update({f"out/{k}": res[k].numpy().astype(np.float32) for k in keep})
    out.update({f"grad/{k}": v.numpy().astype(np.float32) for k, v in grads.items()})
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)


def digest_indices(n, k=512):
    """fixed, evenly spread sample positions of a flattened tensor (all of it when it is small)"""
    return np.arange(n) if n <= k else np.linspace(0, n - 1, k).round().astype(np.int64)


def model_case_full_size(name, cfg, B, gstep, param_seed=11, bias_std=0.15):
    """BASELINE-size fixture (configs[1]: 50x50 / 20x20 / T=3 at batch 64).  The 2.6 M parameters and their gradients would
    be 20 MB as raw arrays, so the fixture holds: the seed the parameters are drawn from plus a float64 checksum per tensor
    (the test regenerates them with O.init_params and verifies the checksums), obs, noise, every per-sample output, the
    scalar losses, and per gradient tensor a 512-point evenly spread sample + its sum and abs-sum in float64."""
    params = O.init_params(cfg, seed=param_seed, bias_std=bias_std)
    obs, _ = O.synthetic_batch(cfg, B, seed=12)
    noise = O.make_noise(cfg, B, seed=13)
    p64 = {k: v.double() for k, v in params.items()}
    res, grads = O.forward_backward(p64, cfg, obs.double(), {k: v.double() for k, v in noise.items()}, global_step=gstep)
    out = {"obs": obs.numpy(), "global_step": np.int64(gstep), "param_seed": np.int64(param_seed),
           "param_bias_std": np.float64(bias_std)}
    out.update({f"param_sum/{k}": np.float64(v.double().sum().item()) for k, v in params.items()})
    out.update({f"param_abs/{k}": np.float64(v.double().abs().sum().item()) for k, v in params.items()})
    out.update({f"noise/{k}": v.numpy() for k, v in noise.items()})
    keep = ["glimpse", "what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob",
            "presence", "final_canvas", "rec_loss_per_sample", "kl_num_steps_per_sample", "kl_what_per_sample",
            "kl_where_per_sample", "num_steps_posterior", "prior_step_weight", "num_steps_log_prob", "baseline",
            "rec_loss", "kl_num_steps", "kl_what", "kl_where", "loss", "reinforce_loss", "baseline_loss", "opt_loss"]
    out.update({f"out/{k}": res[k].numpy().astype(np.float32) for k in keep})
    for k, v in grads.items():
        flat = v.numpy().reshape(-1)
        idx = digest_indices(flat.size)
        out[f"grad_sample/{k}"] = flat[idx].astype(np.float32)
        out[f"grad_sum/{k}"] = np.float64(flat.sum())
        out[f"grad_abs/{k}"] = np.float64(np.abs(flat).sum())
        out[f"grad_max/{k}"] = np.float64(np.abs(flat).max())
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)


def prior_known_answers():
    """The numbers held by the reference's own tests (test/prior_test.py:15-24, 100-120)."""
    np.savez(os.path.join(HERE, "prior_known_answers.npz"),
             geom_prob=np.float64(.75), geom_n=np.int64(10),
             geom_expected=(1. - .75) * .75 ** np.arange(11),
             bern_in=np.array([[0., 0., 0.], [1., 0., 0.], [1., 1., 0.], [1., 1., 1.], [.5, .5, .5]], np.float32),
             bern_out=np.array([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.],
                                [.5, .25, .125, .125]], np.float32))


if __name__ == "__main__":
    st_cases()
    model_case("tiny", O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3,
                                     transform_var_bias=0.5), 6, 20000)
    model_case("small", O.AIRConfig(img_size=(16, 16), crop_size=(6, 6), n_appearance=8, n_hidden=24,
                                    inpt_encoder_hidden=(32, 24), glimpse_encoder_hidden=(28,),
                                    glimpse_decoder_hidden=(20, 16), transform_estimator_hidden=(18,),
                                    steps_pred_hidden=(12, 6), baseline_hidden=(16, 8), max_steps=3), 5, 1500)
    model_case_full_size("c2_b64", O.AIRConfig(), 64, 20000)
    prior_known_answers()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
