"""The torch-CPU oracle (oracle/air_oracle.py) against an independent scalar-loop C coding of the same pieces
(oracle/net_loops.c): affine/ELU, the LSTM step, Gaussian sampling + KL, the num-steps posterior / prior / KL (incl. the
reference's own known answers, test/prior_test.py) and the centred RMSProp update.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import air_oracle as O
from oracle import net_loops as C


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize("elu", [False, True])
def test_affine_matches(dt, tol, elu):
    rng = np.random.default_rng(0)
    x, w, b = rng.normal(size=(7, 33)).astype(dt), (rng.normal(size=(33, 19)) / 6).astype(dt), rng.normal(size=19).astype(dt)
    ref = O.affine(t(x), t(w), t(b), elu=elu).numpy()
    np.testing.assert_allclose(C.affine(x, w, b, elu), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_lstm_step_matches(dt, tol):
    rng = np.random.default_rng(1)
    M, I, H = 5, 11, 9
    x, h, c = (rng.normal(size=s).astype(dt) for s in ((M, I), (M, H), (M, H)))
    w, b = (rng.normal(size=(I + H, 4 * H)) / 4).astype(dt), rng.normal(size=4 * H).astype(dt)
    h2, c2 = O.lstm_step(t(x), t(h), t(c), t(w), t(b), forget_bias=1.0)
    ch, cc = C.lstm_step(x, h, c, w, b, 1.0)
    np.testing.assert_allclose(ch, h2.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(cc, c2.numpy(), rtol=tol, atol=tol)


def test_gauss_sample_and_kl_match():
    rng = np.random.default_rng(2)
    M, D = 6, 50
    pre, eps = rng.normal(size=(M, 2 * D)), rng.normal(size=(M, D))
    loc, scale, sample, kl = C.gauss_sample_kl(pre, eps, 0.5, 0.0, 1.0)
    tl, tr = t(pre)[:, :D], t(pre)[:, D:]
    ts = torch.nn.functional.softplus(tr + 0.5)
    np.testing.assert_allclose(scale, ts.numpy(), rtol=1e-12)
    np.testing.assert_allclose(sample, (tl + ts * t(eps)).numpy(), rtol=1e-12)
    ref = torch.distributions.kl_divergence(torch.distributions.Normal(tl, ts),
                                            torch.distributions.Normal(torch.zeros(()).double(), torch.ones(()).double())).sum(-1)
    np.testing.assert_allclose(kl, ref.numpy(), rtol=1e-11)


def test_numsteps_posterior_known_answers_and_oracle():
    # test/prior_test.py:100-120 -- the reference's own exact vectors
    cases = [([0., 0., 0.], [1., 0., 0., 0.]), ([1., 1., 1.], [0., 0., 0., 1.]), ([.5, .5, .5], [.5, .25, .125, .125]),
             ([1., 0., 0.], [0., 1., 0., 0.]), ([.1, .2, .3], [.9, .08, .014, .006])]
    for p, q in cases:
        np.testing.assert_allclose(C.numsteps_posterior(np.array([p]))[0], q, rtol=1e-12, atol=1e-15)
    rng = np.random.default_rng(3)
    p = rng.uniform(0, 1, size=(40, 5))
    np.testing.assert_allclose(C.numsteps_posterior(p), O.bernoulli_to_modified_geometric(t(p)).numpy(), rtol=1e-12, atol=1e-15)


def test_geometric_prior_and_tabular_kl_match():
    for s in (1. - 1e-15, 0.5, 1e-3, 1e-7, 0.0):
        np.testing.assert_allclose(C.geometric_prior(s, 3), O.geometric_prior(s, 3).numpy(), rtol=1e-12)
    # test/prior_test.py:15-24: Geometric(probs=1-s).prob(k)
    np.testing.assert_allclose(C.geometric_prior(0.3, 3), [0.7 * 0.3 ** k for k in range(4)], rtol=1e-12)
    rng = np.random.default_rng(4)
    q = C.numsteps_posterior(rng.uniform(0, 1, size=(30, 3)))
    q[0] = [0.5, 0.0, 0.0, 0.5]                                           # zeros are skipped, not NaN (prior_test.py:141-204)
    pi = C.geometric_prior(0.2, 3)
    ref = O.tabular_kl(t(q), t(pi)).sum(-1).numpy()
    np.testing.assert_allclose(C.tabular_kl(q, pi), ref, rtol=1e-12)
    assert np.all(np.isfinite(C.tabular_kl(q, pi)))
    np.testing.assert_allclose(C.tabular_kl(np.array([pi / pi.sum()]), pi / pi.sum()), [0.0], atol=1e-15)   # prior_test.py:40-44


def test_rmsprop_centered_matches_oracle_step():
    rng = np.random.default_rng(5)
    cfg = O.tiny_config()
    params = {"w": t(rng.normal(size=(4, 3)))}
    grads = {"w": t(rng.normal(size=(4, 3)))}
    slots = O.rmsprop_init(params)
    p = params["w"].numpy().copy().reshape(-1)
    ms, mg, mom = np.ones_like(p), np.zeros_like(p), np.zeros_like(p)
    for _ in range(3):
        O.rmsprop_centered_step(params, grads, slots, cfg)
        C.rmsprop_centered(p, grads["w"].numpy().reshape(-1), ms, mg, mom, cfg.learning_rate, cfg.rms_decay,
                           cfg.rms_momentum, cfg.rms_eps)
    np.testing.assert_allclose(p, params["w"].numpy().reshape(-1), rtol=1e-12)
