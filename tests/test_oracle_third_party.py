"""The oracle's restatements against PyTorch's own, independently written implementations of the same algorithms (CPU).

These do not pin TF 1.1 / Sonnet 1.1 conventions (gate ORDER, forget-bias placement, eps placement are assumptions listed in
tests/golden/ASSUMPTIONS.md); they pin the algebra: once the convention is mapped, a third party's LSTM cell and centred
RMSProp give the same numbers as oracle/air_oracle.py."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import air_oracle as O


def test_lstm_step_equals_torch_lstmcell_after_gate_permutation():
    torch.manual_seed(0)
    M, I, H = 6, 10, 7
    x, h, c = torch.randn(M, I).double(), torch.randn(M, H).double(), torch.randn(M, H).double()
    w = (torch.randn(I + H, 4 * H) / 4).double()              # Sonnet layout [in+hid, 4*hid], gate order i, j, f, o
    b = torch.randn(4 * H).double()
    h2, c2 = O.lstm_step(x, h, c, w, b, forget_bias=1.0)
    cell = torch.nn.LSTMCell(I, H).double()                    # torch: weight_ih [4H, I], gate order i, f, g, o
    i_, j_, f_, o_ = (slice(k * H, (k + 1) * H) for k in range(4))
    order = [i_, f_, j_, o_]
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([w[:I, s] for s in order], 1).t())
        cell.weight_hh.copy_(torch.cat([w[I:, s] for s in order], 1).t())
        bias = torch.cat([b[s] for s in order]).clone()
        bias[H:2 * H] += 1.0                                   # Sonnet adds forget_bias=1 inside the sigmoid
        cell.bias_ih.copy_(bias)
        cell.bias_hh.zero_()
    th, tc = cell(x, (h, c))
    assert torch.allclose(h2, th, rtol=1e-12, atol=1e-12) and torch.allclose(c2, tc, rtol=1e-12, atol=1e-12)


def test_centered_rmsprop_equals_torch_rmsprop_up_to_eps_placement():
    """TF: mom <- m*mom + lr*g/sqrt(ms - mg^2 + eps), ms initialised to ONE.  torch: buf <- m*buf + g/(sqrt(ms - mg^2) + eps),
    p <- p - lr*buf, ms initialised to zero.  With a constant lr and ms preset to one the trajectories differ only by the
    placement of eps = 1e-10 (relative 1e-10)."""
    torch.manual_seed(1)
    cfg = O.AIRConfig()
    p0 = torch.randn(5, 4).double()
    params = {"input_encoder/0/w": p0.clone()}                 # a model (not baseline) variable: lr multiplier 1
    slots = O.rmsprop_init(params)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([tp], lr=cfg.learning_rate, alpha=cfg.rms_decay, eps=cfg.rms_eps, momentum=cfg.rms_momentum,
                              centered=True)
    for it in range(5):
        g = torch.randn(5, 4, generator=torch.Generator().manual_seed(10 + it)).double()
        O.rmsprop_centered_step(params, {"input_encoder/0/w": g}, slots, cfg)
        tp.grad = g.clone()
        if it == 0:                                            # materialise torch's state, then apply TF's ms = 1 initialisation
            opt.step()
            st = opt.state[tp]
            with torch.no_grad():
                tp.copy_(p0); st["square_avg"].fill_(1.0); st["grad_avg"].zero_(); st["momentum_buffer"].zero_()
                st["step"] = torch.tensor(0.0) if torch.is_tensor(st["step"]) else 0
            tp.grad = g.clone()
        opt.step()
    np.testing.assert_allclose(params["input_encoder/0/w"].numpy(), tp.detach().numpy(), rtol=1e-8, atol=1e-12)


def test_normal_kl_equals_torch_distributions():
    """ASSUMPTIONS #7: TF's _kl_normal_normal in the oracle against torch.distributions' independently written registration."""
    g = torch.Generator().manual_seed(2)
    mu_a, mu_b = torch.randn(7, 5, generator=g).double(), torch.randn(7, 5, generator=g).double()
    s_a, s_b = torch.rand(7, 5, generator=g).double() * 2 + 1e-3, torch.rand(7, 5, generator=g).double() * 2 + 1e-3
    ref = torch.distributions.kl_divergence(torch.distributions.Normal(mu_a, s_a), torch.distributions.Normal(mu_b, s_b))
    assert torch.allclose(O.normal_kl(mu_a, s_a, mu_b, s_b), ref, rtol=1e-12, atol=1e-13)
    # the script's priors: N(0, 1)
    ref0 = torch.distributions.kl_divergence(torch.distributions.Normal(mu_a, s_a), torch.distributions.Normal(0., 1.))
    assert torch.allclose(O.normal_kl(mu_a, s_a, torch.zeros(()).double(), torch.ones(()).double()), ref0, rtol=1e-12, atol=1e-13)


def test_softplus_scale_sample_equals_torch_normal_rsample():
    """ASSUMPTIONS #6: NormalWithSoftplusScale(loc, raw).sample() = loc + softplus(raw) * eps.  torch's Normal.rsample draws
    eps with torch.normal under the global generator: re-seeding reproduces the same eps for the oracle's explicit form."""
    g = torch.Generator().manual_seed(3)
    loc, raw = torch.randn(6, 4, generator=g).double(), torch.randn(6, 4, generator=g).double() * 3
    torch.manual_seed(123)
    ref = torch.distributions.Normal(loc, F.softplus(raw)).rsample()
    torch.manual_seed(123)
    eps = torch.normal(torch.zeros(6, 4).double(), torch.ones(6, 4).double())
    cfg = O.AIRConfig()
    emb = torch.cat([torch.randn(6, 4, generator=g).double(), raw - cfg.transform_var_bias], -1)
    o_loc, o_raw = O.transform_params(emb, cfg)
    assert torch.allclose(o_raw, raw, rtol=0, atol=1e-15)
    got = loc + F.softplus(o_raw) * eps                              # cell.py:130-133 as the oracle writes it
    assert torch.allclose(got, ref, rtol=1e-13, atol=1e-13)
    # softplus itself against its definition log(1 + exp(x)) in a range where that is exact enough
    x = torch.linspace(-30, 30, 121).double()
    # (torch returns x itself above its threshold of 20: exp(-20) = 2e-9 absolute, far below fp32 resolution there)
    assert torch.allclose(F.softplus(x), torch.log1p(torch.exp(x)), rtol=1e-9, atol=1e-300)


def test_geometric_prior_equals_scipy_geom():
    """ASSUMPTIONS #9: Geometric(probs=1-s).prob(k) on k = 0..n (number of failures before the first success).
    scipy.stats.geom counts TRIALS (support 1, 2, ...), so pmf_scipy(k + 1, p) is the same number."""
    from scipy import stats
    for s in (0.75, 1e-5, 0.5, 1.0 - 1e-9):
        ours = O.geometric_prior(s, 10).numpy()
        ref = stats.geom.pmf(np.arange(11) + 1, 1.0 - s)
        np.testing.assert_allclose(ours, ref, rtol=1e-9, atol=1e-300)
    # nbinom(1, p) is the failures-before-first-success form directly
    np.testing.assert_allclose(O.geometric_prior(0.3, 5).numpy(), stats.nbinom.pmf(np.arange(6), 1, 0.7), rtol=1e-12)


def test_elu_affine_equals_definition():
    """neural.py:56-60 with tf.nn.elu: x if x > 0 else exp(x) - 1 (alpha = 1)."""
    g = torch.Generator().manual_seed(4)
    x, w, b = torch.randn(5, 9, generator=g).double(), torch.randn(9, 6, generator=g).double(), torch.randn(6, generator=g).double()
    y = x @ w + b
    ref = torch.where(y > 0, y, torch.expm1(y))
    assert torch.allclose(O.affine(x, w, b, elu=True), ref, rtol=1e-13, atol=1e-15)
    assert torch.allclose(O.affine(x, w, b, elu=False), y, rtol=0, atol=0)


def test_tabular_kl_equals_scipy_rel_entr_and_num_steps_log_prob_equals_categorical():
    from scipy import special
    g = torch.Generator().manual_seed(5)
    p = torch.rand(8, 4, generator=g).double(); p[0, 1] = 0.0; p = p / p.sum(-1, keepdim=True)
    q = torch.rand(8, 4, generator=g).double() + 1e-3
    np.testing.assert_allclose(O.tabular_kl(p, q).numpy(), special.rel_entr(p.numpy(), q.numpy()), rtol=1e-12, atol=1e-300)
    n = torch.randint(0, 4, (8,), generator=g)
    pp = p.clone(); pp[0] = torch.tensor([0.1, 0.2, 0.3, 0.4], dtype=torch.float64)
    ref = torch.distributions.Categorical(probs=pp).log_prob(n)
    got = O.num_steps_log_prob(pp, n)
    ok = pp.gather(1, n.reshape(-1, 1)).reshape(-1) > 0
    assert torch.allclose(got[ok], ref[ok], rtol=1e-9, atol=1e-9)   # Categorical re-normalises and clamps its probs


def test_bernoulli_to_modified_geometric_equals_explicit_product_form():
    """prior.py:62-68 against the textbook form q(n) = (1 - p_{n+1}) prod_{t<=n} p_t, q(T) = prod_t p_t (already normalised)."""
    g = torch.Generator().manual_seed(6)
    p = torch.rand(9, 5, generator=g).double()
    T = p.shape[1]
    ref = torch.zeros(9, T + 1).double()
    for b in range(9):
        run = 1.0
        for n in range(T):
            ref[b, n] = run * (1.0 - p[b, n].item())
            run *= p[b, n].item()
        ref[b, T] = run
    assert torch.allclose(O.bernoulli_to_modified_geometric(p), ref, rtol=1e-12, atol=1e-15)
    assert torch.allclose(ref.sum(-1), torch.ones(9).double(), rtol=1e-12, atol=0)


def test_truncated_normal_init_statistics():
    """ASSUMPTIONS #5: TruncNormal(0, 1/sqrt(fan_in)) re-drawn outside +-2 sigma -- moments of the +-2 sigma truncated normal
    (scipy.stats.truncnorm) and the hard bound."""
    from scipy import stats
    cfg = O.AIRConfig()
    w = O.init_params(cfg, seed=3)["input_encoder/0/w"].double().numpy()
    sigma = 1.0 / np.sqrt(w.shape[0])
    assert np.abs(w).max() <= 2.0 * sigma * (1 + 1e-6)
    ref_std = stats.truncnorm.std(-2.0, 2.0) * sigma
    assert abs(w.std() - ref_std) / ref_std < 5e-3 and abs(w.mean()) < 5e-3 * sigma
