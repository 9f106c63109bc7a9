"""Sanity of the oracle's cell / objective / optimiser restatement: API shapes of test/cell_test.py, fp64 finite
differences of the whole objective, the closed form of the [B,B] NVIL quirk, RMSProp known answer."""
import math

import numpy as np
import torch

from oracle import air_oracle as O


def test_cell_api_shapes_like_reference_smoke():        # test/cell_test.py:21-63
    cfg = O.tiny_config(transition="gru")
    B = 10
    params = O.init_params(cfg, seed=1)
    obs = torch.rand(B, *cfg.img_size)
    noise = O.make_noise(cfg, B)
    state = O.initial_state(params, cfg, obs)
    outs, state = O.cell_step(params, cfg, state, noise["eps_where"][0], noise["eps_what"][0], noise["u_pres"][0])
    widths = [9, 4, 10, 10, 10, 4, 4, 4, 1, 1]           # cell.py:82-95
    assert [o.shape for o in outs] == [(B, n) for n in widths]
    assert len(state) == 6
    res = O.unroll(params, cfg, obs, noise)
    assert res["canvas"].shape == (3, B, 3, 3) and res["glimpse"].shape == (3, B, 2, 2)


def test_write_only_state_and_hoistable_encoder():
    """what/where in the state are never read by the next step (SURVEY B-6)."""
    cfg = O.tiny_config()
    params = O.init_params(cfg, seed=3, bias_std=0.3)
    obs = torch.rand(4, *cfg.img_size); noise = O.make_noise(cfg, 4)
    s = O.initial_state(params, cfg, obs)
    s2 = list(s); s2[2] = torch.randn_like(s[2]); s2[3] = torch.randn_like(s[3])
    a, _ = O.cell_step(params, cfg, s, noise["eps_where"][0], noise["eps_what"][0], noise["u_pres"][0])
    b, _ = O.cell_step(params, cfg, s2, noise["eps_where"][0], noise["eps_what"][0], noise["u_pres"][0])
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_objective_gradients_fd_fp64():
    cfg = O.tiny_config(step_bias=0.3, explore_eps=1e-3, output_multiplier=0.5, output_std=0.3,
                        transform_var_bias=0.5)
    B = 5
    params = O.init_params(cfg, seed=5, dtype=torch.float64, bias_std=0.2)
    obs = torch.rand(B, *cfg.img_size, dtype=torch.float64)
    noise = O.make_noise(cfg, B, seed=7, dtype=torch.float64)
    res, grads = O.forward_backward(params, cfg, obs, noise, global_step=20000)
    rng = np.random.default_rng(0)
    eps = 1e-6
    for name in ["input_encoder/0/w", "lstm/w_gates", "lstm/h0", "transform/1/w", "steps/1/b", "what/w",
                 "glimpse_decoder/1/w", "glimpse_encoder/0/b"]:
        p = params[name]
        for _ in range(3):
            idx = tuple(int(rng.integers(0, s)) for s in p.shape)
            orig = p[idx].item()
            vals = []
            for d in (+eps, -eps):
                p[idx] = orig + d
                r = O.objective(params, cfg, obs, noise, 20000)
                # REINFORCE term: importance weight is a constant wrt params
                vals.append((r["loss"] + (res["importance_weight"] * r["num_steps_log_prob"]).mean()).item())
            p[idx] = orig
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - grads[name][idx].item()) < 1e-5 * max(1.0, abs(fd)), (name, idx, fd, grads[name][idx])
    # baseline gradient
    p = params["baseline/1/w"]; idx = (1, 2); orig = p[idx].item(); vals = []
    for d in (+eps, -eps):
        p[idx] = orig + d
        vals.append(O.objective(params, cfg, obs, noise, 20000)["baseline_loss"].item())
    p[idx] = orig
    assert abs((vals[0] - vals[1]) / (2 * eps) - grads["baseline/1/w"][idx].item()) < 1e-6


def test_nvil_quirk_closed_form():
    """SURVEY B-1: the [B]-[B,1] broadcast makes the learned baseline act as a batch-mean scalar."""
    cfg = O.tiny_config()
    B = 6
    params = O.init_params(cfg, seed=2, bias_std=0.5)
    obs = torch.rand(B, *cfg.img_size); noise = O.make_noise(cfg, B)
    r = O.objective(params, cfg, obs, noise)
    rec, b, lq = r["rec_loss_per_sample"], r["baseline"].reshape(B), r["num_steps_log_prob"]
    assert r["importance_weight"].shape == (B, B)
    closed = ((rec - b.mean()) * lq).mean()
    assert torch.allclose(r["reinforce_loss"], closed, rtol=1e-5, atol=1e-6)
    closed_b = 0.5 * ((rec[None, :] - b[:, None]) ** 2).mean()
    assert torch.allclose(r["baseline_loss"], closed_b, rtol=1e-6)


def test_rec_loss_formula():
    cfg = O.tiny_config(output_std=0.3, output_multiplier=0.5)
    params = O.init_params(cfg, seed=2)
    obs = torch.rand(3, *cfg.img_size); noise = O.make_noise(cfg, 3)
    r = O.objective(params, cfg, obs, noise)
    d = torch.distributions.Normal(r["final_canvas"], 0.3)
    assert torch.allclose(r["rec_loss_per_sample"], -d.log_prob(obs).sum((1, 2)), rtol=1e-5)


def test_rmsprop_centered_known_answer():
    cfg = O.AIRConfig()
    p = {"a/w": torch.tensor([1.0, -2.0])}
    g = {"a/w": torch.tensor([0.5, -0.25])}
    s = O.rmsprop_init(p)
    O.rmsprop_centered_step(p, g, s, cfg)
    ms = 0.9 + 0.1 * np.array([0.25, 0.0625]); mg = 0.1 * np.array([0.5, -0.25])
    mom = 1e-4 * np.array([0.5, -0.25]) / np.sqrt(ms - mg ** 2 + 1e-10)
    np.testing.assert_allclose(p["a/w"].numpy(), np.array([1.0, -2.0]) - mom, rtol=1e-6)
    p2 = {"baseline/0/w": torch.tensor([1.0])}; s2 = O.rmsprop_init(p2)
    O.rmsprop_centered_step(p2, {"baseline/0/w": torch.tensor([0.5])}, s2, cfg)
    assert abs((1.0 - p2["baseline/0/w"].item()) - 10 * mom[0]) < 1e-7      # baseline lr x10, model.py:363


def test_param_counts_match_survey_appendix_d():
    cfg = O.AIRConfig()
    shapes = O.param_shapes(cfg)
    model = sum(int(np.prod(s)) for k, s in shapes.items() if not O.is_baseline_param(k))
    base = sum(int(np.prod(s)) for k, s in shapes.items() if O.is_baseline_param(k))
    assert model == 1_782_525 and base == 846_593            # incl. trainable (h0,c0)
    assert cfg.baseline_in == 3177
