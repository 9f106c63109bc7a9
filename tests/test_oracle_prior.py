"""The oracle's number-of-steps prior math against every known answer of the reference's test/prior_test.py
(the only reference tests that pin hot-path numbers)."""
import numpy as np
import torch
from numpy.testing import assert_array_almost_equal, assert_array_equal

from oracle import air_oracle as O


def test_geometric_prior_known_answer():           # test/prior_test.py:15-24
    prob, n_steps = 0.75, 10
    expected = (1.0 - prob) * prob ** np.arange(n_steps + 1)
    p = O.geometric_prior(prob, n_steps).numpy()
    assert_array_almost_equal(p, expected)


def test_tabular_kl_same():                        # test/prior_test.py:40-44
    p = torch.tensor([[0.25] * 4], dtype=torch.float32)
    kl = O.tabular_kl(p, p).numpy()
    assert kl.shape == (1, 4)
    assert kl.sum() == 0.0


def test_tabular_kl_zero():                        # test/prior_test.py:46-52
    p = torch.tensor([[0.0, 0.25, 0.25, 0.5]]); q = torch.tensor([[0.25] * 4])
    kl = O.tabular_kl(p, q).numpy()
    assert kl.sum() > 0 and np.isfinite(kl).all()


def test_tabular_kl_one():                         # test/prior_test.py:54-60
    p = torch.tensor([[0.0, 1.0, 0.0, 0.0]]); q = torch.tensor([[1.0 - 1e-7, 1e-7, 0.0, 0.0]])
    kl = O.tabular_kl(p, q).numpy()
    assert kl.sum() > 0 and np.isfinite(kl).all()


def test_tabular_kl_positive_on_random():          # test/prior_test.py:62-74
    rng = np.random.default_rng(0)
    for _ in range(100):
        a = np.abs(rng.random((1, 4))); a /= a.sum()
        b = np.abs(rng.random((1, 4))); b /= b.sum()
        kl = O.tabular_kl(torch.tensor(a, dtype=torch.float32), torch.tensor(b, dtype=torch.float32)).numpy()
        assert kl.sum() > 0


def test_modified_geometric_shapes():              # test/prior_test.py:86-98
    for shape in [(3,), (7, 3), (7, 11, 3)]:
        out = O.bernoulli_to_modified_geometric(torch.rand(*shape))
        assert tuple(out.shape) == shape[:-1] + (4,)


def test_modified_geometric_obvious():             # test/prior_test.py:100-115
    cases = {(0., 0., 0.): [1., 0., 0., 0.], (1., 0., 0.): [0., 1., 0., 0.],
             (1., 1., 0.): [0., 0., 1., 0.], (1., 1., 1.): [0., 0., 0., 1.]}
    for p, exp in cases.items():
        assert_array_equal(O.bernoulli_to_modified_geometric(torch.tensor(p)).numpy(), exp)


def test_modified_geometric_geom():                # test/prior_test.py:117-120
    out = O.bernoulli_to_modified_geometric(torch.tensor([0.5, 0.5, 0.5])).numpy()
    assert_array_equal(out, [0.5, 0.5 ** 2, 0.5 ** 3, 0.5 ** 3])


def test_num_steps_kl_stress_finite():             # test/prior_test.py:141-186
    prior = O.geometric_prior(0.005, 3)
    rng = np.random.default_rng(1)
    for _ in range(100):
        p = torch.tensor(rng.random((1, 3)), dtype=torch.float32, requires_grad=True)
        kl = O.tabular_kl(O.bernoulli_to_modified_geometric(p), prior[None])
        assert kl.sum() > 0 and torch.isfinite(kl).all()
        g, = torch.autograd.grad(kl.sum(), p)
        assert torch.isfinite(g).all()
        free = torch.tensor(rng.random((1, 4)), dtype=torch.float32); free = (free / free.sum()).requires_grad_(True)
        fkl = O.tabular_kl(free, prior[None])
        g2, = torch.autograd.grad(fkl.sum(), free)
        assert fkl.sum() > 0 and torch.isfinite(fkl).all() and torch.isfinite(g2).all()


def test_posterior_zeros():                        # test/prior_test.py:188-204
    prior = O.geometric_prior(0.005, 3)
    p = torch.tensor([[0.5, 0.0, 0.0]], requires_grad=True)
    post = O.bernoulli_to_modified_geometric(p)
    kl = O.tabular_kl(post, prior[None])
    assert kl.sum() > 0 and torch.isfinite(kl).all()
    g, = torch.autograd.grad(kl.sum(), p)
    assert torch.isfinite(g).all()


def test_anneal_schedule_matches_closed_form():    # model.py:106-124 with multi_mnist.py:40-47
    cfg = O.AIRConfig()
    assert O.steps_prior_success_prob(cfg, 0) == cfg.nsp_init
    assert O.steps_prior_success_prob(cfg, 1000) == cfg.nsp_init
    s = O.steps_prior_success_prob(cfg, 51000)
    expected = (1 - 1e-15) * ((1e-7 / (1 - 1e-15)) ** (1e4 / 1e5)) ** (50000 / 1e4)
    assert abs(s - expected) < 1e-12
    assert O.steps_prior_success_prob(cfg, 10 ** 7) == cfg.nsp_final
