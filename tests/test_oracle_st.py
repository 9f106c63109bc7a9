"""Cross-checks of the oracle's spatial transformer: three independent codings (torch gather, torch grid_sample,
C scalar loops with analytic gradients), fp64 finite differences and the hand-derivable cases of SURVEY 8c."""
import numpy as np
import pytest
import torch

from oracle import air_oracle as O
from oracle import st_loops as C


def _rand_where(B, rng, wide=False):
    sx = rng.uniform(0.2, 1.4, B) * (rng.choice([-1, 1], B) if wide else 1)
    sy = rng.uniform(0.2, 1.4, B)
    tx = rng.uniform(-0.8, 0.8, B); ty = rng.uniform(-0.8, 0.8, B)
    return np.stack([sx, tx, sy, ty], 1)


@pytest.mark.parametrize("H,W,h,w", [(50, 50, 20, 20), (7, 5, 3, 4), (100, 100, 28, 28)])
def test_read_three_codings_agree(H, W, h, w):
    rng = np.random.default_rng(0)
    B = 6
    img = rng.random((B, H, W)).astype(np.float32)
    where = _rand_where(B, rng, wide=True).astype(np.float32)
    a = O.st_read(torch.tensor(img), torch.tensor(where), (h, w)).numpy()
    b = O.st_read_gridsample(torch.tensor(img), torch.tensor(where), (h, w)).numpy()
    c = C.st_read_fwd(img, where, (h, w))
    np.testing.assert_array_equal(a, c)                    # same op order -> bit-exact
    np.testing.assert_allclose(a, b, atol=2e-5)            # grid_sample orders its arithmetic differently


@pytest.mark.parametrize("H,W,h,w", [(50, 50, 20, 20), (7, 5, 3, 4)])
def test_write_three_codings_agree(H, W, h, w):
    rng = np.random.default_rng(1)
    B = 6
    glm = rng.standard_normal((B, h, w)).astype(np.float32)
    where = _rand_where(B, rng, wide=True).astype(np.float32)
    a = O.st_write(torch.tensor(glm), torch.tensor(where), (H, W)).numpy()
    b = O.st_write_gridsample(torch.tensor(glm), torch.tensor(where), (H, W)).numpy()
    c = C.st_write_fwd(glm, where, (H, W))
    np.testing.assert_array_equal(a, c)
    np.testing.assert_allclose(a, b, atol=5e-5)


def test_identity_transform():
    """where=[1,0,1,0] with equal in/out size is the identity (SURVEY 8c iii)."""
    img = torch.rand(3, 9, 11, dtype=torch.float64)
    where = torch.tensor([[1., 0., 1., 0.]] * 3, dtype=torch.float64)
    np.testing.assert_allclose(O.st_read(img, where, (9, 11)).numpy(), img.numpy(), atol=1e-12)
    np.testing.assert_allclose(O.st_write(img, where, (9, 11)).numpy(), img.numpy(), atol=1e-12)


def test_glimpse_outside_is_zero():
    img = torch.rand(2, 10, 10)
    where = torch.tensor([[0.1, 5.0, 0.1, 0.0], [0.1, 0.0, 0.1, -7.0]])
    assert O.st_read(img, where, (4, 4)).abs().max() == 0
    assert (C.st_read_fwd(img.numpy(), where.numpy(), (4, 4)) == 0).all()


def test_write_of_constant_inside_box():
    """write of a constant glimpse paints the constant strictly inside the attended box, zero strictly outside."""
    g = torch.ones(1, 6, 6, dtype=torch.float64) * 3.0
    where = torch.tensor([[0.5, 0.0, 0.5, 0.0]], dtype=torch.float64)
    out = O.st_write(g, where, (21, 21))[0]
    X = torch.linspace(-1, 1, 21, dtype=torch.float64)
    inside = (X.abs() < 0.5 - 1e-9)
    assert torch.allclose(out[inside][:, inside], torch.full((int(inside.sum()),) * 2, 3.0, dtype=torch.float64))
    far = (X.abs() > 0.5 + 2.0 / 5 * 0.5 + 1e-9)         # one glimpse pixel beyond the edge
    assert out[far].abs().max() == 0 and out[:, far].abs().max() == 0


def test_read_gradients_autograd_vs_analytic_vs_fd():
    rng = np.random.default_rng(2)
    B, H, W, h, w = 4, 9, 8, 5, 6
    img = rng.random((B, H, W)); where = _rand_where(B, rng); dout = rng.standard_normal((B, h, w))
    ti = torch.tensor(img, requires_grad=True); tw = torch.tensor(where, requires_grad=True)
    out = O.st_read(ti, tw, (h, w))
    gi, gw = torch.autograd.grad((out * torch.tensor(dout)).sum(), [ti, tw])
    dwhere, dimg = C.st_read_bwd(img, where, dout)
    np.testing.assert_allclose(gw.numpy(), dwhere, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gi.numpy(), dimg, rtol=1e-10, atol=1e-10)
    # finite differences wrt where (piecewise-smooth: random points are interior w.p. 1)
    eps = 1e-6
    for k in range(4):
        wp, wm = where.copy(), where.copy(); wp[:, k] += eps; wm[:, k] -= eps
        fd = ((C.st_read_fwd(img, wp, (h, w)) - C.st_read_fwd(img, wm, (h, w))) * dout).sum((1, 2)) / (2 * eps)
        np.testing.assert_allclose(fd, dwhere[:, k], rtol=1e-5, atol=1e-6)


def test_write_gradients_autograd_vs_analytic_vs_fd():
    rng = np.random.default_rng(3)
    B, H, W, h, w = 4, 9, 8, 5, 6
    glm = rng.standard_normal((B, h, w)); where = _rand_where(B, rng, wide=True); dout = rng.standard_normal((B, H, W))
    tg = torch.tensor(glm, requires_grad=True); tw = torch.tensor(where, requires_grad=True)
    out = O.st_write(tg, tw, (H, W))
    gg, gw = torch.autograd.grad((out * torch.tensor(dout)).sum(), [tg, tw])
    dglm, dwhere = C.st_write_bwd(glm, where, dout)
    np.testing.assert_allclose(gg.numpy(), dglm, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(gw.numpy(), dwhere, rtol=1e-9, atol=1e-9)
    eps = 1e-7
    for k in range(4):
        wp, wm = where.copy(), where.copy(); wp[:, k] += eps; wm[:, k] -= eps
        fd = ((C.st_write_fwd(glm, wp, (H, W)) - C.st_write_fwd(glm, wm, (H, W))) * dout).sum((1, 2)) / (2 * eps)
        np.testing.assert_allclose(fd, dwhere[:, k], rtol=2e-4, atol=1e-5)


def test_negative_and_tiny_scale_do_not_crash():
    """sx is a *sample* and may be <= 0 (SURVEY B-11): the inverse uses 1/sx; inf/nan coordinates sample zero."""
    g = torch.rand(3, 4, 4)
    where = torch.tensor([[-0.5, 0.1, 0.7, 0.0], [0.0, 0.0, 1.0, 0.0], [1e-30, 0.3, 1.0, 0.0]])
    out = O.st_write(g, where, (8, 8))
    c = C.st_write_fwd(g.numpy(), where.numpy(), (8, 8))
    assert torch.isfinite(out[0]).all()
    np.testing.assert_array_equal(np.nan_to_num(out.numpy(), nan=12345.0), np.nan_to_num(c, nan=12345.0))
