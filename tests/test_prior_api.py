"""The product's prior.py / ops.py (host-side, device-agnostic utilities) against every known answer of the reference's
test/prior_test.py -- the reference's own unit tests, ported line by line (runs on CPU)."""
import numpy as np
import torch
from numpy.testing import assert_array_almost_equal, assert_array_equal

from attend_infer_repeat_amd.ops import Loss, clip_preserve
from attend_infer_repeat_amd.prior import (NumStepsDistribution, bernoulli_to_modified_geometric, geometric_prior,
                                           sample_from_tensor, tabular_kl)

_N_STRESS_ITER = 100


def test_geometric_prior():                                   # test/prior_test.py:13-24
    prob, n_steps = .75, 10
    expected = (1. - prob) * prob ** np.arange(n_steps + 1)
    assert_array_almost_equal(geometric_prior(prob, n_steps).numpy(), expected)
    assert geometric_prior(torch.tensor(.75), 10).dtype == torch.float32


def test_tabular_kl_same():                                   # :40-44
    p = torch.tensor([[.25] * 4])
    kl = tabular_kl(p, p, 0.).numpy()
    assert kl.shape == (1, 4) and kl.sum() == 0.


def test_tabular_kl_zero_and_one():                           # :46-60
    kl = tabular_kl(torch.tensor([[0., .25, .25, .5]]), torch.tensor([[.25] * 4]), 0.)
    assert kl.sum() > 0
    kl = tabular_kl(torch.tensor([[0., 1., 0., 0.]]), torch.tensor([[1. - 1e-7, 1e-7, 0., 0.]]), 0.)
    assert kl.sum() > 0 and torch.isfinite(kl).all()


def test_tabular_kl_positive_on_random():                     # :62-74
    rng = np.random.default_rng(0)
    for _ in range(_N_STRESS_ITER):
        p = np.abs(rng.random((1, 4))); p /= p.sum()
        q = np.abs(rng.random((1, 4))); q /= q.sum()
        assert tabular_kl(torch.tensor(p, dtype=torch.float32), torch.tensor(q, dtype=torch.float32)).sum() > 0


def test_modified_geometric_shape_and_values():               # :86-120
    for shape in [(3,), (7, 3), (7, 11, 3)]:
        assert tuple(bernoulli_to_modified_geometric(torch.rand(*shape)).shape) == shape[:-1] + (4,)
    for p, e in {(0., 0., 0.): [1., 0., 0., 0.], (1., 0., 0.): [0., 1., 0., 0.], (1., 1., 0.): [0., 0., 1., 0.],
                 (1., 1., 1.): [0., 0., 0., 1.]}.items():
        assert_array_equal(bernoulli_to_modified_geometric(torch.tensor(p)).numpy(), e)
    assert_array_equal(bernoulli_to_modified_geometric(torch.tensor([.5, .5, .5])).numpy(), [.5, .5 ** 2, .5 ** 3, .5 ** 3])


def test_num_steps_kl_stress_and_zeros():                     # :141-204
    prior = geometric_prior(.005, 3)
    rng = np.random.default_rng(1)
    for _ in range(_N_STRESS_ITER):
        p = torch.tensor(rng.random((1, 3)), dtype=torch.float32, requires_grad=True)
        kl = tabular_kl(bernoulli_to_modified_geometric(p), prior, 0.)
        g, = torch.autograd.grad(kl.sum(), p)
        assert kl.sum() > 0 and torch.isfinite(kl).all() and torch.isfinite(g).all()
    p = torch.tensor([[.5, 0., 0.]], requires_grad=True)
    post = bernoulli_to_modified_geometric(p)
    gpost, = torch.autograd.grad(post.sum(), p, retain_graph=True)
    kl = tabular_kl(post, prior, 0.)
    g, = torch.autograd.grad(kl.sum(), p)
    assert kl.sum() > 0 and torch.isfinite(kl).all() and torch.isfinite(g).all() and torch.isfinite(gpost).all()


def test_num_steps_distribution_log_prob_and_gather():       # prior.py:93-151
    probs = torch.tensor([[.9, .5, .1], [.2, .3, .4]])
    d = NumStepsDistribution(probs)
    q = d.prob()
    assert torch.allclose(q.sum(-1), torch.ones(2))
    idx = torch.tensor([2., 0.])
    assert torch.allclose(d.prob(idx), torch.stack([q[0, 2], q[1, 0]]))
    assert torch.allclose(sample_from_tensor(q, idx), d.prob(idx))
    assert torch.allclose(d.log_prob(idx), torch.log(d.prob(idx)))
    s = d.sample()
    assert s.shape == (2,) and (s >= 0).all() and (s <= 3).all()


def test_clip_preserve_and_loss_helper():                     # ops.py:5-43,67-76
    x = torch.tensor([1e-40, 0.5], requires_grad=True)
    y = clip_preserve(x, 1e-32, x.detach())
    assert y[0].item() >= 1e-32 and y[1].item() == 0.5
    g, = torch.autograd.grad(y.sum(), x)
    assert torch.equal(g, torch.ones(2))                      # chain rule preserved
    a, b = Loss(), Loss()
    a.add(torch.tensor(1.), torch.ones(3)); b.add(torch.tensor(2.), 2 * torch.ones(3), weight=.5)
    a.add(b, weight=2.)
    assert a.value.item() == 3. and torch.equal(a.per_sample, torch.full((3,), 3.))
