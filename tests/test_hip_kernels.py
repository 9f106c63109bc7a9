"""GPU parity of every C-ABI kernel against the CPU oracle on seeded inputs (run with -m gpu on an MI355X).

Tolerances (fp32): forward ST is bit-exact (same op order, explicitly rounded); everything that reduces in a different
order is checked to rtol 2e-5..1e-4 against an fp64 evaluation of the oracle, scaled by the magnitude of the terms."""
import numpy as np
import pytest
import torch

from oracle import air_oracle as O
from oracle import st_loops as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(gpu_device):
    from attend_infer_repeat_amd import hip as H
    H.lib()
    return H


def g(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def rand_where(B, rng, wide=False):
    sx = rng.uniform(0.2, 1.4, B) * (rng.choice([-1, 1], B) if wide else 1)
    sy = rng.uniform(0.2, 1.4, B)
    return np.stack([sx, rng.uniform(-0.8, 0.8, B), sy, rng.uniform(-0.8, 0.8, B)], 1).astype(np.float32)


def assert_close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (tol {tol.flat[err.argmax()]:.3e}) at {np.unravel_index(err.argmax(), err.shape)}"


# ---------------------------------------------------------------------------------------------------------------
# spatial transformer
# ---------------------------------------------------------------------------------------------------------------
ST_SHAPES = [(50, 50, 20, 20), (100, 100, 28, 28), (7, 5, 3, 4), (3, 3, 2, 2), (9, 11, 9, 11),
             # narrow images whose pixel count is a multiple of 4 (16-byte groups cross several row boundaries: ADVICE r04)
             (8, 1, 3, 2), (4, 2, 2, 2), (4, 3, 2, 3), (12, 1, 4, 1)]


@pytest.mark.parametrize("H,W,h,w", ST_SHAPES)
def test_st_read_fwd_bit_exact(hip, H, W, h, w):
    rng = np.random.default_rng(0)
    B = 9
    img = rng.random((B, H, W)).astype(np.float32)
    where = rand_where(B, rng, wide=True)
    ref = C.st_read_fwd(img, where, (h, w))
    out = hip.st_read_fwd(g(img), g(where), (h, w)).cpu().numpy()
    np.testing.assert_array_equal(out, ref)


def test_st_read_fwd_batched_over_steps(hip):
    """T glimpses per image (row k reads image k % n_img): the engine's batched unroll."""
    rng = np.random.default_rng(1)
    B, T, H, W, h, w = 5, 3, 50, 50, 20, 20
    img = rng.random((B, H, W)).astype(np.float32)
    where = rand_where(T * B, rng)
    ref = C.st_read_fwd(np.tile(img, (T, 1, 1)), where, (h, w))
    out = hip.st_read_fwd(g(img), g(where), (h, w)).cpu().numpy()
    np.testing.assert_array_equal(out, ref)


def test_st_read_identity_and_outside(hip):
    img = torch.rand(3, 9, 11)
    ident = torch.tensor([[1., 0., 1., 0.]] * 3)
    out = hip.st_read_fwd(img.cuda(), ident.cuda(), (9, 11)).cpu()
    assert torch.allclose(out, img, atol=1e-6)
    far = torch.tensor([[0.1, 5.0, 0.1, 0.0]] * 3)
    assert hip.st_read_fwd(img.cuda(), far.cuda(), (4, 4)).abs().max().item() == 0.0


@pytest.mark.parametrize("H,W,h,w", ST_SHAPES[:4])
def test_st_read_bwd(hip, H, W, h, w):
    rng = np.random.default_rng(2)
    B = 7
    img = rng.random((B, H, W)).astype(np.float32)
    where = rand_where(B, rng)
    dout = rng.standard_normal((B, h, w)).astype(np.float32)
    dwhere64, dimg64 = C.st_read_bwd(img.astype(np.float64), where.astype(np.float64), dout.astype(np.float64))
    dwhere, dimg = hip.st_read_bwd(g(img), g(where), g(dout), want_dimg=True)
    scale = np.abs(dout).sum((1, 2))[:, None] * max(H, W) / 2 * 0.05 + 1.0
    assert_close(dwhere.cpu().numpy() / scale, dwhere64 / scale, 1e-4, 2e-5, "dwhere")
    assert_close(dimg, dimg64, 1e-4, 1e-5, "dimg")
    dwhere2, none = hip.st_read_bwd(g(img), g(where), g(dout), want_dimg=False)
    assert none is None
    assert_close(dwhere2, dwhere, 1e-6, 1e-6, "dwhere w/o dimg")


# (21, 19, 6, 10): h*w a multiple of 4 but w not -- 16-byte groups would straddle the rows of the bordered LDS copy: scalar staging
@pytest.mark.parametrize("H,W,h,w", ST_SHAPES[:4] + [(21, 19, 6, 10)])
def test_st_write_fwd_bit_exact(hip, H, W, h, w):
    rng = np.random.default_rng(3)
    B = 8
    glm = rng.standard_normal((B, h, w)).astype(np.float32)
    where = rand_where(B, rng, wide=True)
    pres = rng.integers(0, 2, B).astype(np.float32)
    canvas = rng.standard_normal((B, H, W)).astype(np.float32)
    inv = C.st_write_fwd(glm, where, (H, W))
    ref = canvas + pres[:, None, None] * inv
    out = hip.st_write_fwd(g(glm), g(where), (H, W), presence=g(pres), canvas_in=g(canvas)).cpu().numpy()
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out0 = hip.st_write_fwd(g(glm), g(where), (H, W)).cpu().numpy()          # no presence, zero canvas
    np.testing.assert_array_equal(out0, inv)


def test_st_write_degenerate_scales(hip):
    glm = torch.rand(3, 4, 4)
    where = torch.tensor([[-0.5, 0.1, 0.7, 0.0], [0.0, 0.0, 1.0, 0.0], [1e-30, 0.3, 1.0, 0.0]])
    ref = C.st_write_fwd(glm.numpy(), where.numpy(), (8, 8))
    out = hip.st_write_fwd(glm.cuda(), where.cuda(), (8, 8)).cpu().numpy()
    np.testing.assert_array_equal(np.nan_to_num(out, nan=12345.0), np.nan_to_num(ref, nan=12345.0))


@pytest.mark.parametrize("H,W,h,w", ST_SHAPES[:4] + [(21, 19, 6, 10)])
def test_st_write_bwd(hip, H, W, h, w):
    rng = np.random.default_rng(4)
    B = 6
    glm = rng.standard_normal((B, h, w)).astype(np.float32)
    where = rand_where(B, rng, wide=True)
    pres = rng.uniform(0.2, 1.0, B).astype(np.float32)
    dout = rng.standard_normal((B, H, W)).astype(np.float32)
    tg = torch.tensor(glm, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(where, dtype=torch.float64, requires_grad=True)
    tp = torch.tensor(pres, dtype=torch.float64, requires_grad=True)
    out = tp[:, None, None] * O.st_write(tg, tw, (H, W))
    gg, gw, gp = torch.autograd.grad((out * torch.tensor(dout, dtype=torch.float64)).sum(), [tg, tw, tp])
    dg, dwhere, dpres = hip.st_write_bwd(g(glm), g(where), g(dout), presence=g(pres), want_dpresence=True)
    assert_close(dg, gg, 1e-4, 1e-4, "dglimpse")
    scale = gw.abs().max(1, keepdim=True)[0].numpy() + 1.0
    assert_close(dwhere.cpu().numpy() / scale, gw.numpy() / scale, 2e-4, 2e-5, "dwhere")
    assert_close(dpres, gp, 1e-4, 1e-3, "dpresence")


def test_canvas_unroll_fwd_bwd(hip):
    rng = np.random.default_rng(5)
    T, B, H, W, h, w = 3, 6, 50, 50, 20, 20
    glm = rng.standard_normal((T, B, h, w)).astype(np.float32)
    where = rand_where(T * B, rng).reshape(T, B, 4)
    pres = np.cumprod(rng.integers(0, 2, (T, B)), 0).astype(np.float32)
    pres[:, 0] = 1.0
    obs = rng.random((B, H, W)).astype(np.float32)
    mult, std = 0.5, 0.3
    # oracle: sequential accumulation
    canvas = np.zeros((B, H, W), np.float32)
    steps = []
    for t in range(T):
        canvas = canvas + pres[t][:, None, None] * C.st_write_fwd(glm[t], where[t], (H, W))
        steps.append(canvas.copy())
    st, final, rec = hip.canvas_unroll_fwd(g(glm), g(where), g(pres), (H, W), obs=g(obs), mult=mult, std=std)
    np.testing.assert_array_equal(st.cpu().numpy(), np.stack(steps))
    np.testing.assert_array_equal(final.cpu().numpy(), steps[-1])
    tg = torch.tensor(glm, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(where, dtype=torch.float64, requires_grad=True)
    cv = sum(torch.tensor(pres[t], dtype=torch.float64)[:, None, None] * O.st_write(tg[t], tw[t], (H, W)) for t in range(T))
    nll = 0.5 * ((torch.tensor(obs, dtype=torch.float64) - mult * cv) / std) ** 2 + 0.5 * np.log(2 * np.pi) + np.log(std)
    rec64 = nll.sum((1, 2))
    assert_close(rec, rec64, 1e-5, 1e-3, "rec_per_sample")
    gg, gw = torch.autograd.grad(rec64.mean(), [tg, tw])
    dg, dwhere = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, mult, std, 1.0 / B)
    assert_close(dg, gg, 2e-4, 1e-4 * gg.abs().max().item(), "dglimpse")
    scale = gw.abs().amax(-1, keepdim=True).numpy() + 1.0
    assert_close(dwhere.cpu().numpy() / scale, gw.numpy() / scale, 5e-4, 5e-5, "dwhere")
    # the recompute form (no final canvas: each unit re-forms the canvas on its footprint) is BITWISE the same backward
    dg2, dwhere2 = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), None, mult, std, 1.0 / B)
    assert torch.equal(dg, dg2) and torch.equal(dwhere, dwhere2)


@pytest.mark.parametrize("T,B,H,W,h,w", [(5, 3, 100, 100, 28, 28), (1, 4, 17, 13, 5, 7), (4, 70, 28, 36, 9, 12), (3, 1100, 50, 50, 20, 20),
                                          (2, 5, 21, 19, 6, 10)])
def test_canvas_unroll_bwd_recompute_equals_stored_canvas(hip, T, B, H, W, h, w):
    """air_canvas_unroll_bwd(final_canvas=NULL) against the form that reads the stored final canvas, bit for bit, on the
    BASELINE configs[3] shapes, odd sizes (scalar staging paths), negative / degenerate scales and a grid-strided batch."""
    rng = np.random.default_rng(T * 1000 + B)
    glm = rng.standard_normal((T, B, h, w)).astype(np.float32)
    where = rand_where(T * B, rng).reshape(T, B, 4)
    where[0, 0] = [-0.7, 0.2, 0.9, -0.1]                                   # mirrored glimpse
    if B > 2:
        where[-1, 2] = [3.0, 0.0, 3.0, 0.0]                                # glimpse larger than the canvas
    pres = np.cumprod(rng.integers(0, 2, (T, B)), 0).astype(np.float32); pres[:, 0] = 1.0
    obs = rng.random((B, H, W)).astype(np.float32)
    _, final, _ = hip.canvas_unroll_fwd(g(glm), g(where), g(pres), (H, W), obs=g(obs), mult=0.5, std=0.3, keep_steps=False)
    dg, dwhere = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, 0.5, 0.3, 1.0 / B)
    dg2, dwhere2 = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), None, 0.5, 0.3, 1.0 / B)
    assert torch.equal(dg, dg2) and torch.equal(dwhere, dwhere2)


@pytest.mark.parametrize("T,B,H,W,h,w", [(3, 1100, 50, 50, 20, 20), (5, 40, 100, 100, 28, 28), (1, 300, 17, 13, 5, 7), (4, 600, 28, 36, 9, 12),
                                          (3, 200, 50, 50, 20, 20), (2, 9, 70, 130, 24, 70)])
def test_canvas_kernels_in_every_workgroup_shape_match_the_oracle(hip, T, B, H, W, h, w):
    """The row-streaming canvas kernels pick their workgroup shape from the number of units (many waves per unit while the launch
    does not fill the chip, one wave per unit beyond; canvases wider than 64 columns and glimpses wider than 64 take several lane
    chunks): per-step canvases and the final canvas BIT for bit the oracle's sequential accumulation in every shape, the
    reconstruction term, dglimpse and dwhere against float64 autograd -- incl. mirrored / oversized / off-canvas glimpses, absent
    steps, odd sizes and grid-strided batches.  dglimpse does not depend on the shape at all (every element is accumulated by
    one lane in row order): the latency-regime and throughput-regime launches of the same problem agree bit for bit."""
    rng = np.random.default_rng(T * 100 + B)
    glm = rng.standard_normal((T, B, h, w)).astype(np.float32)
    where = rand_where(T * B, rng).reshape(T, B, 4)
    where[0, 0] = [-0.7, 0.2, 0.9, -0.13]        # mirrored (not -0.1: canvas row 0 would land EXACTLY on glimpse row 0, where the
                                                 # fp32 and fp64 coordinates fall on different sides of the cell boundary and d/dy jumps)
    if B > 2:
        where[-1, 2] = [3.0, 0.0, 3.0, 0.0]
        where[0, 1] = [0.3, 5.0, 0.3, 0.0]                                  # entirely outside the canvas
    pres = np.cumprod(rng.integers(0, 2, (T, B)), 0).astype(np.float32); pres[:, 0] = 1.0
    obs = rng.random((B, H, W)).astype(np.float32)
    mult, std = 0.5, 0.3
    st, final, rec = hip.canvas_unroll_fwd(g(glm), g(where), g(pres), (H, W), obs=g(obs), mult=mult, std=std)
    nb_chk = min(B, 48)                                                     # oracle on a slice of the batch (it is a CPU loop)
    canvas = np.zeros((nb_chk, H, W), np.float32)
    for t in range(T):
        canvas = canvas + pres[t, :nb_chk][:, None, None] * C.st_write_fwd(glm[t, :nb_chk], where[t, :nb_chk], (H, W))
        np.testing.assert_array_equal(st[t, :nb_chk].cpu().numpy(), canvas)
    np.testing.assert_array_equal(final[:nb_chk].cpu().numpy(), canvas)
    # gradients against the oracle evaluated in FLOAT32: d/dwhere jumps at cell boundaries, and with tens of thousands of sampling
    # coordinates per case some land within an fp32 rounding of an integer -- the fp64 oracle then differentiates another cell.
    # (The forward check above proves the kernel's coordinates, floors and weights ARE the fp32 oracle's.)
    tg = torch.tensor(glm[:, :nb_chk], requires_grad=True)
    tw = torch.tensor(where[:, :nb_chk], requires_grad=True)
    cv = sum(torch.tensor(pres[t, :nb_chk])[:, None, None] * O.st_write(tg[t], tw[t], (H, W)) for t in range(T))
    nll = 0.5 * ((torch.tensor(obs[:nb_chk]) - mult * cv) / std) ** 2 + 0.5 * np.log(2 * np.pi) + np.log(std)
    rec32 = nll.double().sum((1, 2))
    assert_close(rec[:nb_chk], rec32, 1e-5, 1e-3, "rec_per_sample")
    gg, gw = torch.autograd.grad(nll.sum() / B, [tg, tw])
    dg, dwhere = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, mult, std, 1.0 / B)
    assert_close(dg[:, :nb_chk], gg, 5e-4, 2e-4 * gg.abs().max().item(), "dglimpse")
    scale = gw.abs().amax(-1, keepdim=True).numpy() + 1.0 / B
    assert_close(dwhere[:, :nb_chk].cpu().numpy() / scale, gw.numpy() / scale, 2e-3, 2e-4, "dwhere")
    # the same images as a small batch (another workgroup shape): forward and dglimpse bit for bit, dwhere / rec to rounding
    nb2 = min(B, 5)
    sl = lambda a: g(np.ascontiguousarray(a[:, :nb2]))
    st2, final2, rec2 = hip.canvas_unroll_fwd(sl(glm), sl(where), sl(pres), (H, W), obs=g(obs[:nb2]), mult=mult, std=std)
    assert torch.equal(st2, st[:, :nb2]) and torch.equal(final2, final[:nb2])
    assert_close(rec2, rec[:nb2], 1e-5, 1e-3, "rec")
    # (beyond 2048 units the stored-canvas backward runs image-major -- st_write_bwd_gs_kernel<false, true>: one workgroup per image,
    #  dcanvas formed once, presence applied to the finished element -- which agrees with the unit-major form to rounding)
    img_major = T * B > 2048 and T <= 8
    for fc in (final2, None):
        dg2, dwhere2 = hip.canvas_unroll_bwd(sl(glm), sl(where), sl(pres), g(obs[:nb2]), fc, mult, std, 1.0 / B)
        if img_major:
            assert_close(dg2, dg[:, :nb2], 2e-5, 2e-6 * float(dg.abs().max()), "dglimpse")
        else:
            assert torch.equal(dg2, dg[:, :nb2])
        assert_close(dwhere2, dwhere[:, :nb2], 2e-4, 1e-5 * float(dwhere.abs().max()) + 1e-7, "dwhere")


@pytest.mark.parametrize("T,B,H,W,h,w", [(3, 704, 50, 50, 20, 20), (5, 420, 40, 60, 12, 16), (1, 2100, 17, 13, 5, 7), (2, 1100, 50, 50, 21, 19)])
def test_canvas_backward_image_major_equals_unit_major(hip, monkeypatch, T, B, H, W, h, w):
    """Rounds 5-6: beyond 2048 units the stored-canvas backward runs one workgroup per IMAGE (image-major form of st_write_bwd_gs_kernel:
    dcanvas formed once per image, contraction weights once per glimpse column / row, 4 barriers per image) instead of one per (t, b)
    unit.  Same tables, exact ranges and dwhere chain: dglimpse / dwhere agree with the unit-major form to rounding on ordinary,
    mirrored, tiny, oversized (ranges wider than four canvas columns: the general loop) and off-canvas transforms, and are NaN /
    inf exactly where it is on degenerate scales (the reference's inverse warp has no guard: tests/test_extreme_scales.py)."""
    rng = np.random.default_rng(T * 1000 + B)
    glm = rng.standard_normal((T, B, h, w)).astype(np.float32)
    where = rand_where(T * B, rng, wide=True).reshape(T, B, 4)
    where[0, 0] = [-0.7, 0.2, 0.9, -0.13]
    where[-1, 1] = [3.0, 0.0, 3.0, 0.0]
    where[0, 2] = [0.3, 5.0, 0.3, 0.0]
    where[-1, 3] = [1.3, 0.05, 0.95, -0.02]           # every glimpse column touched by 6-7 canvas columns
    where[0, 4] = [0.04, 0.1, 0.05, -0.3]             # a glimpse smaller than two canvas pixels
    for i, v in enumerate([0.0, -0.0, 1e-40, 1e-30, 1e-20, -1e-20, 1e19, 3e38]):
        where[i % T, 8 + i] = [v, 0.3, 0.7, -0.2]
        where[(i + 1) % T, 20 + i] = [0.6, -0.1, v, 0.25]
    pres = np.cumprod(rng.integers(0, 2, (T, B)), 0).astype(np.float32); pres[:, :40] = 1.0
    obs = rng.random((B, H, W)).astype(np.float32)
    mult, std = 0.5, 0.3
    _, final, _ = hip.canvas_unroll_fwd(g(glm), g(where), g(pres), (H, W), obs=g(obs), mult=mult, std=std)
    monkeypatch.setenv("AIR_CANVAS_BWD_IMG", "0")
    dg_u, dw_u = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, mult, std, 1.0 / B)
    monkeypatch.setenv("AIR_CANVAS_BWD_IMG", "1")
    dg_i, dw_i = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, mult, std, 1.0 / B)
    torch.cuda.synchronize()
    for name, a, b in (("dglimpse", dg_i, dg_u), ("dwhere", dw_i, dw_u)):
        a, b = a.cpu(), b.cpu()
        cls = lambda t: torch.isnan(t) * 3 + torch.isposinf(t) * 1 + torch.isneginf(t) * 2
        assert torch.equal(cls(a), cls(b)), name + ": non-finite placement differs"
        fin = torch.isfinite(b)
        if name == "dglimpse":
            err = ((a - b)[fin].abs().max() / b[fin].abs().max()).item()
            assert err < 5e-6, (name, err)
        else:
            # rows: relative to the row's largest component (the four components share their partial sums)
            af, bf = torch.where(fin, a, torch.zeros_like(a)), torch.where(fin, b, torch.zeros_like(b))
            scale = bf.abs().amax(-1, keepdim=True) + 1e-6 * bf.abs().max() + 1e-30
            err = ((af - bf).abs() / scale).max().item()
            assert err < 2e-3, (name, err)
    # (with a binary presence the two forms' dglimpse is the same FMA chain -- they differ in the sign of the zeros an absent step
    #  leaves, which is how one can tell that the switch took effect)


# ---------------------------------------------------------------------------------------------------------------
# GEMM / linear / LSTM
# ---------------------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(64, 256, 2500), (192, 256, 400), (10, 5, 9), (64, 8, 256), (192, 1, 64), (64, 100, 256),
               (64, 400, 256), (3, 17, 3), (64, 256, 3177), (192, 256, 50), (33, 47, 129), (2048, 256, 400),
               (3072, 100, 256), (1000, 130, 70), (3177, 256, 1024)]          # the last three: large-batch shapes


def _gemm_ref(A, B, ta, tb):
    a = A.double().cpu(); b = B.double().cpu()
    return (a.t() if ta else a) @ (b.t() if tb else b)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_all_layouts(hip, M, N, K, ta, tb):
    gen = torch.Generator().manual_seed(M * 131 + N * 17 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=gen).cuda()
    B = torch.randn((N, K) if tb else (K, N), generator=gen).cuda()
    ref = _gemm_ref(A, B, ta, tb)
    out = hip.gemm(A, B, ta=bool(ta), tb=bool(tb))
    assert_close(out, ref, 1e-5, 2e-6 * K ** 0.5 * 4, f"gemm {M}x{N}x{K} ta={ta} tb={tb}")
    out2 = hip.gemm(A, B, ta=bool(ta), tb=bool(tb), use_workspace=False)       # no split-K path
    assert_close(out2, ref, 1e-5, 2e-6 * K ** 0.5 * 4, "gemm no-workspace")


@pytest.mark.parametrize("precision", [0, 1])
def test_gemm_grouped_throughput_regime(hip, precision):
    """One launch in the throughput regime (thousands of 16x16 tiles -> 32x32 tiles per wave): the dW / dX pair of a layer at
    batch 1024, an odd-sized problem with unaligned leading dimensions, every epilogue, colsum."""
    gen = torch.Generator().manual_seed(5 + precision)
    r = (lambda t: t.to(torch.bfloat16).double()) if precision else (lambda t: t.double())
    M, K, N = 3072, 256, 400
    x = torch.randn(M, K, generator=gen).cuda(); g = torch.randn(M, N, generator=gen).cuda()
    w = (torch.randn(K, N, generator=gen) / 16).cuda(); y = torch.randn(M, K, generator=gen).cuda()
    big = torch.randn(1001, 77, generator=gen).cuda(); a2 = big[:, 5:75]                 # [1001, 70], ld 77, unaligned
    b2 = torch.randn(70, 131, generator=gen).cuda(); bias2 = torch.randn(131, generator=gen).cuda()
    outs = hip.gemm_grouped([
        dict(A=x, B=g, ta=True, colsum=True),                                            # dW[K,N] = x^T g, db
        dict(A=g, B=w, tb=True, epilogue=hip.EPI_MUL_DELU, aux=y),                       # dX[M,K] = (g w^T) * elu'(y)
        dict(A=a2, B=b2, bias=bias2, epilogue=hip.EPI_BIAS_ELU),
        dict(A=b2, B=a2, ta=True, tb=True),                                              # [131, 1001] = b2^T a2^T
    ], precision=precision)
    tol = dict(rtol=1e-5, atol=3e-4) if not precision else dict(rtol=1e-5, atol=3e-4)
    xc, gc, wc, yc = x.cpu(), g.cpu(), w.cpu(), y.cpu()
    assert_close(outs[0][0], r(xc).t() @ r(gc), tol["rtol"], tol["atol"] * 10, "dW")
    assert_close(outs[0][1], gc.double().sum(0), 1e-5, 1e-3, "db")
    d = torch.where(yc > 0, torch.ones_like(yc), yc + 1).double()
    assert_close(outs[1][0], (r(gc) @ r(wc).t()) * d, tol["rtol"], tol["atol"], "dX")
    ref2 = r(a2.cpu()) @ r(b2.cpu())
    assert_close(outs[2][0], torch.nn.functional.elu(ref2 + bias2.cpu().double()), 1e-5, 3e-4, "odd bias+elu")
    assert_close(outs[3][0], ref2.t(), 1e-5, 3e-4, "TT")


def test_gemm_transpose_detecting_identity(hip):
    """A = I with an asymmetric B catches row/col swaps in the C write."""
    B = torch.arange(48 * 40, dtype=torch.float32).reshape(48, 40).cuda()
    out = hip.gemm(torch.eye(48).cuda(), B)
    assert torch.equal(out, B)


def test_gemm_epilogues_views_beta_colsum(hip):
    gen = torch.Generator().manual_seed(7)
    M, N, K = 70, 45, 133
    big = torch.randn(M, K + 11, generator=gen).cuda()
    A = big[:, 3:3 + K]                                   # row view with a leading dimension, unaligned start
    Bm = torch.randn(K, N, generator=gen).cuda()
    bias = torch.randn(N, generator=gen).cuda()
    aux = torch.randn(M, N, generator=gen).cuda()
    ref = A.double().cpu() @ Bm.double().cpu()
    assert_close(hip.gemm(A, Bm, bias=bias, epilogue=hip.EPI_BIAS), ref + bias.double().cpu(), 1e-5, 1e-4, "bias")
    elu = torch.nn.functional.elu(ref + bias.double().cpu())
    assert_close(hip.gemm(A, Bm, bias=bias, epilogue=hip.EPI_BIAS_ELU), elu, 1e-5, 1e-4, "bias+elu")
    y = aux
    d = torch.where(y > 0, torch.ones_like(y), y + 1).double().cpu()
    assert_close(hip.gemm(A, Bm, aux=aux, epilogue=hip.EPI_MUL_DELU), ref * d, 1e-5, 1e-4, "mul delu")
    assert_close(hip.gemm(A, Bm, aux=aux, bias=bias, epilogue=hip.EPI_ADD_AUX), ref + aux.double().cpu() + bias.double().cpu(),
                 1e-5, 1e-4, "add aux")
    C0 = torch.randn(M, N, generator=gen).cuda()
    out = hip.gemm(A, Bm, beta=1.0, out=C0.clone())
    assert_close(out, ref + C0.double().cpu(), 1e-5, 1e-4, "beta")
    X = torch.randn(64, 300, generator=gen).cuda(); G = torch.randn(64, 40, generator=gen).cuda()
    dw, cs = hip.gemm(X, G, ta=True, colsum=True)
    assert_close(dw, X.double().cpu().t() @ G.double().cpu(), 1e-5, 1e-4, "dW")
    assert_close(cs, G.double().cpu().sum(0), 1e-5, 1e-5, "colsum")


@pytest.mark.parametrize("M,K,N", [(64, 2500, 256), (192, 400, 256), (10, 9, 5), (192, 64, 1), (64, 50, 256)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fwd_bwd(hip, M, K, N, act):
    gen = torch.Generator().manual_seed(M + K + N + act)
    x = torch.randn(M, K, generator=gen); w = torch.randn(K, N, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen); dy = torch.randn(M, N, generator=gen)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    y64 = O.affine(x64, w64, b64, elu=bool(act))
    gx, gw, gb = torch.autograd.grad((y64 * dy.double()).sum(), [x64, w64, b64])
    y = hip.linear_fwd(x.cuda(), w.cuda(), b.cuda(), act)
    assert_close(y, y64, 2e-5, 2e-5, "linear fwd")
    dx, dw, db = hip.linear_bwd(x.cuda(), w.cuda(), y, dy.cuda(), act)
    assert_close(dx, gx, 1e-4, 1e-4, "dx"); assert_close(dw, gw, 1e-4, 2e-4, "dw"); assert_close(db, gb, 1e-4, 1e-4, "db")


def test_lstm_pointwise(hip):
    gen = torch.Generator().manual_seed(11)
    M, Hd = 37, 50
    gates = torch.randn(M, 4 * Hd, generator=gen); c0 = torch.randn(M, Hd, generator=gen)
    dh = torch.randn(M, Hd, generator=gen); dc = torch.randn(M, Hd, generator=gen)
    g64 = gates.double().requires_grad_(True); c64 = c0.double().requires_grad_(True)
    i, j, f, o = torch.chunk(g64, 4, -1)
    c2 = torch.sigmoid(f + 1.0) * c64 + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    gg, gc = torch.autograd.grad((h2 * dh.double()).sum() + (c2 * dc.double()).sum(), [g64, c64])
    h, c, act = hip.lstm_pointwise_fwd(gates.cuda(), c0.cuda(), 1.0)
    assert_close(h, h2, 1e-5, 1e-6, "h"); assert_close(c, c2, 1e-5, 1e-6, "c")
    dgates, dc_prev = hip.lstm_pointwise_bwd(act, c0.cuda(), c, dh.cuda(), dc.cuda())
    assert_close(dgates, gg, 1e-4, 1e-5, "dgates"); assert_close(dc_prev, gc, 1e-4, 1e-5, "dc_prev")


@pytest.mark.parametrize("M,Hd", [(64, 256), (37, 50), (5, 7), (130, 40), (1024, 256), (2005, 64), (1024, 200)])
def test_lstm_fused_step_matches_unfused_pair_and_fp64(hip, M, Hd):
    """air_lstm_step_fwd / _bwd = recurrent product + gate math in one launch; must equal GEMM + pointwise (bitwise: same
    K order, same reduction tree) and the fp64 formula."""
    gen = torch.Generator().manual_seed(M * 7 + Hd)
    h0 = torch.randn(M, Hd, generator=gen) * 0.5; c0 = torch.randn(M, Hd, generator=gen)
    w_full = torch.randn(Hd + 3, 4 * Hd, generator=gen) / Hd ** 0.5
    gx = torch.randn(M, 4 * Hd, generator=gen)
    w_h_dev = w_full.cuda()[3:]                                     # row-offset view, like the engine's w_gates[E:]
    h, c, act = hip.lstm_step_fwd(h0.cuda(), c0.cuda(), w_h_dev, gx.cuda(), 1.0)
    g64 = h0.double() @ w_full[3:].double() + gx.double()
    i, j, f, o = torch.chunk(g64, 4, -1)
    c2 = torch.sigmoid(f + 1.0) * c0.double() + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    assert_close(h, h2, 2e-5, 2e-5, "h"); assert_close(c, c2, 2e-5, 2e-5, "c")
    gates = hip.gemm(h0.cuda(), w_h_dev, epilogue=hip.EPI_ADD_AUX, aux=gx.cuda())
    hu, cu, actu = hip.lstm_pointwise_fwd(gates, c0.cuda(), 1.0)
    assert_close(h, hu, 1e-5, 1e-5, "h vs unfused"); assert_close(act, actu, 1e-5, 1e-5, "act vs unfused")

    # backward link: dh = dgn . W_h^T + dh_a + dh_b, then the pointwise backward of (act, c0, c)
    dgn = torch.randn(M, 4 * Hd, generator=gen) * 0.3
    dh_a = torch.randn(M, Hd, generator=gen); dh_b = torch.randn(M, Hd, generator=gen); dc = torch.randn(M, Hd, generator=gen)
    dgx_in = torch.randn(M, 4 * Hd, generator=gen)
    w_h_c = w_h_dev.contiguous()
    dg, dcp, dgx = hip.lstm_step_bwd(dgn.cuda(), w_h_c, dh_a.cuda(), dh_b.cuda(), dc.cuda(), act, c0.cuda(), c,
                                     dgx_in=dgx_in.cuda(), want_dgx=True)
    dh_tot = dgn.double() @ w_full[3:].double().t() + dh_a.double() + dh_b.double()
    a64 = act.cpu().double(); gi, gj, gf, go = torch.chunk(a64, 4, -1)
    tc = torch.tanh(c.cpu().double())
    dct = dc.double() + dh_tot * go * (1 - tc * tc)
    ref = torch.cat([dct * gj * gi * (1 - gi), dct * gi * (1 - gj * gj), dct * c0.double() * gf * (1 - gf),
                     dh_tot * tc * go * (1 - go)], -1)
    assert_close(dg, ref, 1e-4, 2e-5, "dgates"); assert_close(dcp, dct * gf, 1e-4, 2e-5, "dc_prev")
    assert_close(dgx, dgx_in.double() + ref, 1e-4, 2e-5, "dgx")
    # NULL optional terms
    dg2, dcp2, none = hip.lstm_step_bwd(dgn.cuda(), w_h_c, None, None, None, act, c0.cuda(), c)
    dh0 = dgn.double() @ w_full[3:].double().t()
    assert none is None
    assert_close(dcp2, dh0 * go * (1 - tc * tc) * gf, 1e-4, 2e-5, "dc_prev (no dh/dc terms)")


@pytest.mark.parametrize("shapes", [
    [(10000, 256, 64, True)],                                  # the closing launch's problem (configs[3])
    [(5000, 128, 48, False), (4100, 256, 64, True)],            # two streaming problems, one with three chunks and half the columns
    [(10000, 256, 64, True), (50, 256, 192, True), (787, 256, 64, False)],      # configs[3]'s pixels beside ordinary tile problems
    [(4099, 64, 16, True), (192, 400, 256, False), (4203, 192, 32, False)],     # ragged rows, one chunk, mixed with a dX-shaped product
    [(2500, 256, 64, True)],                                   # below the row threshold: the tile kernels (configs[1])
])
def test_short_k_weight_gradients_on_the_streaming_body(hip, shapes):
    """air_gemm_grouped: TN weight gradients with K <= 64 rows and at least 4096 output rows run on the streaming body
    (shortk_dw_body: the whole dY operand in registers, 16-row slabs grid-stride, no K split); every problem of the group -- streaming
    or tiled -- must match the fp64 product and column sums, with offset row views as the engine passes them."""
    gen = torch.Generator().manual_seed(sum(m + n + k for m, n, k, _ in shapes))
    probs, refs = [], []
    for i, (M, N, K, cs) in enumerate(shapes):
        tn = (M, N, K) != (192, 400, 256)
        if tn:
            A = torch.randn(K, M + 8, generator=gen).cuda()[:, 4:4 + M]          # [K, M] view with a leading dimension
            B = torch.randn(K, N, generator=gen).cuda()
            out = torch.zeros(M + 3, N, device="cuda")[1:1 + M]
            probs.append(dict(A=A, B=B, ta=True, out=out, colsum=cs))
            refs.append((A.double().t() @ B.double(), B.double().sum(0) if cs else None))
        else:
            A = torch.randn(M, K, generator=gen).cuda(); B = torch.randn(N, K, generator=gen).cuda()
            probs.append(dict(A=A, B=B, tb=True))
            refs.append((A.double() @ B.double().t(), None))
    outs = hip.gemm_grouped(probs)
    for (C, cs), (rC, rcs), sh in zip(outs, refs, shapes):
        assert_close(C, rC, 1e-5, 2e-5, "C %s" % (sh,))
        if rcs is not None:
            assert_close(cs, rcs, 1e-5, 2e-5, "colsum %s" % (sh,))


@pytest.mark.parametrize("M,Hd", [(64, 256), (37, 48), (130, 32), (272, 256), (5, 16), (64, 512)])
def test_lstm_bwd_entry_matches_pointwise_plus_link_and_fp64(hip, M, Hd):
    """air_lstm_step_bwd_entry = the pointwise backward of the last step (no dc flowing in) + the BPTT link of the step before it in
    ONE launch (the link's A operand dgates_{T-1} is formed inside every workgroup); must equal the two launches and the fp64 formula,
    with and without an optimiser slice riding along."""
    gen = torch.Generator().manual_seed(M * 13 + Hd)
    rn = lambda *sh: torch.randn(*sh, generator=gen)
    w_h = (rn(Hd, 4 * Hd) / Hd ** 0.5).cuda()
    act1 = torch.sigmoid(rn(M, 4 * Hd)).cuda(); act0 = torch.sigmoid(rn(M, 4 * Hd)).cuda()
    act1[:, Hd:2 * Hd] = torch.tanh(rn(M, Hd)).cuda(); act0[:, Hd:2 * Hd] = torch.tanh(rn(M, Hd)).cuda()
    c0, c1, c2 = rn(M, Hd).cuda(), rn(M, Hd).cuda(), rn(M, Hd).cuda()          # c_seq[T-2], c_seq[T-1], c_seq[T]
    dh_a1, dh_b1, dh_a0, dh_b0 = (rn(M, Hd).cuda() for _ in range(4))
    # the two launches
    import ctypes
    L = hip.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    dg1_ref = torch.empty_like(act1); dcp1_ref = torch.empty_like(c1)
    assert L.air_lstm_pointwise_bwd(P(act1), P(c1), P(c2), P(dh_a1), P(dh_b1), None, P(dg1_ref), P(dcp1_ref), M, Hd, None) == 0
    dg0_ref, dcp0_ref, dgx_ref = hip.lstm_step_bwd(dg1_ref, w_h, dh_a0, dh_b0, dcp1_ref, act0, c0, c1, dgx_in=dg1_ref, want_dgx=True)
    # one launch
    dg1, dcp1, dg0, dcp0, dgx = hip.lstm_step_bwd_entry(act1, c1, c2, dh_a1, dh_b1, w_h, dh_a0, dh_b0, act0, c0, c1)
    assert_close(dg1, dg1_ref, 1e-6, 1e-6, "dgates of the last step"); assert_close(dcp1, dcp1_ref, 1e-6, 1e-6, "dc of the last step")
    assert_close(dg0, dg0_ref, 1e-5, 1e-5, "dgates"); assert_close(dcp0, dcp0_ref, 1e-5, 1e-5, "dc_prev")
    assert_close(dgx, dgx_ref, 1e-5, 1e-5, "dgx")
    # fp64
    a1 = act1.cpu().double(); gi, gj, gf, go = torch.chunk(a1, 4, -1)
    tc = torch.tanh(c2.cpu().double()); dh1 = dh_a1.cpu().double() + dh_b1.cpu().double()
    dct1 = dh1 * go * (1 - tc * tc)
    d1 = torch.cat([dct1 * gj * gi * (1 - gi), dct1 * gi * (1 - gj * gj), dct1 * c1.cpu().double() * gf * (1 - gf), dh1 * tc * go * (1 - go)], -1)
    dh0 = d1 @ w_h.cpu().double().t() + dh_a0.cpu().double() + dh_b0.cpu().double()
    a0 = act0.cpu().double(); gi0, gj0, gf0, go0 = torch.chunk(a0, 4, -1)
    tc0 = torch.tanh(c1.cpu().double())
    dct0 = dct1 * gf + dh0 * go0 * (1 - tc0 * tc0)
    d0 = torch.cat([dct0 * gj0 * gi0 * (1 - gi0), dct0 * gi0 * (1 - gj0 * gj0), dct0 * c0.cpu().double() * gf0 * (1 - gf0),
                    dh0 * tc0 * go0 * (1 - go0)], -1)
    assert_close(dg1, d1, 1e-4, 2e-5, "dgates of the last step vs fp64"); assert_close(dg0, d0, 1e-4, 2e-5, "dgates vs fp64")
    assert_close(dcp0, dct0 * gf0, 1e-4, 2e-5, "dc_prev vs fp64"); assert_close(dgx, d1 + d0, 1e-4, 2e-5, "dgx vs fp64")
    # one direct term only, no running sum; an optimiser slice riding along leaves the launch's own results unchanged
    n, lo, hi = 8192, 1024, 6140
    pbuf, g = rn(n).cuda(), rn(n).cuda()
    ms, mg, mom = torch.rand(n, generator=gen).cuda() + 1.0, rn(n).cuda() * 0.1, rn(n).cuda() * 0.01
    lr = torch.tensor([1e-3]).cuda()
    orig = [t.clone() for t in (pbuf, ms, mg, mom)]
    ref = [t.clone() for t in (pbuf, ms, mg, mom)]
    hip.rmsprop_centered_(ref[0][lo:hi], g[lo:hi], ref[1][lo:hi], ref[2][lo:hi], ref[3][lo:hi], lr)
    sl = hip.rmsprop_slice(pbuf, g, ms, mg, mom, lo, hi, n, lr)
    r = hip.lstm_step_bwd_entry(act1, c1, c2, dh_a1, None, w_h, None, dh_b0, act0, c0, c1, opt=sl, want_dgx=False)
    q = hip.lstm_step_bwd_entry(act1, c1, c2, dh_a1, None, w_h, None, dh_b0, act0, c0, c1, want_dgx=False)
    torch.cuda.synchronize()
    for a, b in zip(r[:4], q[:4]):
        assert torch.equal(a, b)
    assert r[4] is None
    for b_, r_, o_ in zip((pbuf, ms, mg, mom), ref, orig):
        assert torch.equal(b_[:lo], o_[:lo]) and torch.equal(b_[hi:], o_[hi:])
        assert_close(b_[lo:hi], r_[lo:hi], 1e-6, 1e-7, "slice riding on the entry launch")
        assert not torch.equal(b_[lo:hi], o_[lo:hi])


@pytest.mark.parametrize("M,Hd", [(48, 64), (1045, 128)])          # the second: wide-tile form (> 512 tiles), ragged rows
def test_lstm_fused_step_bf16(hip, M, Hd):
    gen = torch.Generator().manual_seed(3)
    r16 = lambda t: t.to(torch.bfloat16).double()
    h0 = torch.randn(M, Hd, generator=gen); c0 = torch.randn(M, Hd, generator=gen)
    w = torch.randn(Hd, 4 * Hd, generator=gen) / 8; gx = torch.randn(M, 4 * Hd, generator=gen)
    h, c, act = hip.lstm_step_fwd(h0.cuda(), c0.cuda(), w.cuda(), gx.cuda(), 1.0, precision=1)
    g64 = r16(h0) @ r16(w) + gx.double()
    i, j, f, o = torch.chunk(g64, 4, -1)
    c2 = torch.sigmoid(f + 1.0) * c0.double() + torch.sigmoid(i) * torch.tanh(j)
    assert_close(c, c2, 2e-5, 2e-5, "c (bf16 operands)")
    dgn = torch.randn(M, 4 * Hd, generator=gen)
    dg, dcp, _ = hip.lstm_step_bwd(dgn.cuda(), w.cuda(), None, None, None, act, c0.cuda(), c, precision=1)
    dh0 = r16(dgn) @ r16(w).t()
    a64 = act.cpu().double(); gi, gj, gf, go = torch.chunk(a64, 4, -1)
    tc = torch.tanh(c.cpu().double())
    assert_close(dcp, dh0 * go * (1 - tc * tc) * gf, 1e-4, 2e-5, "dc_prev (bf16 operands)")


# ---------------------------------------------------------------------------------------------------------------
# stochastic nodes + objective
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,loc_mode,off,prior4", [(50, 0, 0.5, (0., 1., 0., 1.)), (4, 1, 0.5, (0.3, 1.5, -0.2, 0.7)),
                                                    (10, 0, -2.0, (0., 1., 0., 1.))])
def test_gauss_sample_fwd_bwd(hip, D, loc_mode, off, prior4):
    gen = torch.Generator().manual_seed(D)
    M = 41
    wide = torch.randn(M, 2 * D + 3, generator=gen)
    pre = wide[:, :2 * D]
    eps = torch.randn(M, D, generator=gen); dsample = torch.randn(M, D, generator=gen); dkl = torch.randn(M, generator=gen)
    p64 = pre.double().requires_grad_(True)
    loc = p64[:, :D]
    if loc_mode == 1:
        idx = torch.arange(D)
        loc = torch.where(idx % 2 == 1, torch.tanh(loc), torch.sigmoid(loc))
    scale = torch.nn.functional.softplus(p64[:, D:] + off)
    sample = loc + scale * eps.double()
    pl = torch.tensor([prior4[0] if d % 2 == 0 else prior4[2] for d in range(D)], dtype=torch.float64)
    ps = torch.tensor([prior4[1] if d % 2 == 0 else prior4[3] for d in range(D)], dtype=torch.float64)
    kl = O.normal_kl(loc, scale, pl, ps).sum(-1)
    gp, = torch.autograd.grad((sample * dsample.double()).sum() + (kl * dkl.double()).sum(), [p64])
    pre_gpu = wide.cuda()[:, :2 * D]
    l, s, smp, k = hip.gauss_sample_fwd(pre_gpu, eps.cuda(), off, loc_mode, prior4)
    assert_close(l, loc, 1e-5, 1e-6, "loc"); assert_close(s, scale, 1e-5, 1e-6, "scale")
    assert_close(smp, sample, 1e-5, 1e-5, "sample"); assert_close(k, kl, 1e-5, 1e-4, "kl")
    dpre = hip.gauss_sample_bwd(pre_gpu, eps.cuda(), off, loc_mode, prior4, l, s, dsample.cuda(), dkl.cuda())
    assert_close(dpre, gp, 1e-4, 1e-5, "dpre")


def test_presence(hip):
    gen = torch.Generator().manual_seed(3)
    T, B = 3, 300
    logit = torch.randn(T, B, generator=gen); u = torch.rand(T, B, generator=gen)
    p = 1e-3 / 2 + (1 - 1e-3) * torch.sigmoid(logit.double() + 0.75)
    z = (u.double() < p).double()
    pres = torch.cumprod(z, 0)
    prob, pr = hip.presence_fwd(logit.cuda(), u.cuda(), 0.75, 1e-3, True)
    assert_close(prob, p, 1e-6, 1e-7, "prob")
    agree = (pr.cpu().double() == pres).float().mean().item()
    assert agree > 0.995                                                     # u within 1 ulp of p may flip
    dprob = torch.randn(T, B, generator=gen)
    l64 = logit.double().requires_grad_(True)
    p64 = 1e-3 / 2 + (1 - 1e-3) * torch.sigmoid(l64 + 0.75)
    gl, = torch.autograd.grad((p64 * dprob.double()).sum(), [l64])
    assert_close(hip.presence_bwd(logit.cuda(), 0.75, 1e-3, True, dprob.cuda()), gl, 1e-4, 1e-6, "dlogit")
    prob2, pr2 = hip.presence_fwd(logit.cuda(), None, 0.0, None, False)      # non-discrete: presence = prob
    assert torch.equal(prob2, pr2)
    assert_close(prob2, torch.sigmoid(logit.double()), 1e-6, 1e-7, "prob (no eps)")


def test_rec_loglik(hip):
    gen = torch.Generator().manual_seed(5)
    B, P = 9, 2500
    obs = torch.rand(B, 50, 50, generator=gen); canvas = torch.randn(B, 50, 50, generator=gen)
    c64 = canvas.double().requires_grad_(True)
    nll = (0.5 * ((obs.double() - 0.5 * c64) / 0.3) ** 2 + 0.5 * np.log(2 * np.pi) + np.log(0.3)).sum((1, 2))
    assert_close(hip.rec_loglik_fwd(obs.cuda(), canvas.cuda(), 0.5, 0.3), nll, 1e-5, 1e-3, "rec")
    gc, = torch.autograd.grad(nll.mean(), [c64])
    assert_close(hip.rec_loglik_bwd(obs.cuda(), canvas.cuda(), 0.5, 0.3, None, 1.0 / B), gc, 1e-5, 1e-6, "dcanvas")


@pytest.mark.parametrize("T", [3, 5, 1])
def test_numsteps_fwd_bwd(hip, T):
    gen = torch.Generator().manual_seed(T)
    B = 130
    prob = torch.rand(T, B, generator=gen) * 0.98 + 0.01
    prob[:, 0] = 1.0 - 1e-4; prob[:, 1] = 1e-4
    pres = torch.cumprod((torch.rand(T, B, generator=gen) < prob).float(), 0)
    prior = O.geometric_prior(0.3, T)
    p64 = prob.double().requires_grad_(True)
    # oracle path (f32 posterior re-cast to f64 inside tabular_kl, like the reference)
    pf = prob.clone().requires_grad_(True)
    q = O.bernoulli_to_modified_geometric(pf.t())
    kl = O.tabular_kl(q, prior[None]).sum(1)
    w = torch.flip(torch.cumsum(torch.flip(q[:, 1:].t(), [0]), 0), [0])
    logp = O.num_steps_log_prob(q, pres.sum(0))
    qg, klg, logpg, wg = hip.numsteps_fwd(prob.cuda(), pres.cuda(), prior.cuda())
    assert_close(qg, q, 1e-6, 1e-8, "q"); assert_close(klg, kl, 1e-5, 1e-6, "kl")
    assert_close(wg, w, 1e-6, 1e-7, "w"); assert_close(logpg, logp, 1e-5, 1e-6, "logp")
    dw = torch.randn(T, B, generator=gen); dl = torch.randn(B, generator=gen)
    # fp64 reference gradient
    q64 = O.bernoulli_to_modified_geometric(p64.t())
    L = 0.37 * O.tabular_kl(q64, prior[None]).sum() \
        + (torch.flip(torch.cumsum(torch.flip(q64[:, 1:].t(), [0]), 0), [0]) * dw.double()).sum() \
        + (O.num_steps_log_prob(q64, pres.sum(0)) * dl.double()).sum()
    gp, = torch.autograd.grad(L, [p64])
    dprob = hip.numsteps_bwd(prob.cuda(), pres.cuda(), prior.cuda(), 0.37, dw.cuda(), dl.cuda())
    assert_close(dprob, gp, 2e-4, 2e-4, "dprob")


def test_nvil(hip):
    gen = torch.Generator().manual_seed(9)
    B = 64
    imp = torch.rand(B, generator=gen) * 3000 + 500; base = torch.randn(B, generator=gen) * 10
    logp = -torch.rand(B, generator=gen) * 3
    b64 = base.double().requires_grad_(True); l64 = logp.double().requires_grad_(True)
    iw = imp.double()[None, :] - b64[:, None]                                   # [B,B]: (i,j) = imp_j - b_i
    rl = (iw.detach() * l64).mean(); bl = 0.5 * (iw ** 2).mean()
    gl, = torch.autograd.grad(rl, [l64]); gb, = torch.autograd.grad(bl, [b64])
    out, dlogp, dbase = hip.nvil(imp.cuda(), base.cuda(), logp.cuda())
    ref = torch.stack([rl.detach(), bl.detach(), iw.mean().detach(), iw.var(unbiased=False).detach()])
    assert_close(out, ref, 1e-5, 1e-3, "nvil scalars")
    assert_close(dlogp, gl, 1e-5, 1e-5, "dlogp"); assert_close(dbase, gb, 1e-5, 1e-5, "dbaseline")


def test_baseline_pack(hip):
    cfg = O.AIRConfig()
    T, B = 3, 5
    gen = torch.Generator().manual_seed(2)
    obs = torch.rand(B, 50, 50, generator=gen); what = torch.randn(T, B, 50, generator=gen)
    where = torch.randn(T, B, 4, generator=gen); pres = torch.rand(T, B, 1, generator=gen)
    h = torch.randn(B, 256, generator=gen); c = torch.randn(B, 256, generator=gen)
    parts = [t.permute(1, 0, 2).reshape(B, -1) for t in (what, where, pres)] + [h, c]
    ref = torch.cat([obs.reshape(B, -1)] + parts, -1)
    out = hip.baseline_pack(obs.cuda(), what.cuda(), where.cuda(), pres.cuda(), [h.cuda(), c.cuda()])
    assert out.shape == (B, cfg.baseline_in)
    assert torch.equal(out.cpu(), ref)


def test_rmsprop_centered(hip):
    cfg = O.AIRConfig()
    gen = torch.Generator().manual_seed(4)
    n = 10_007
    p = {"a/w": torch.randn(n, generator=gen)}; gr = {"a/w": torch.randn(n, generator=gen)}
    slots = O.rmsprop_init(p)
    pg, gg = p["a/w"].clone().cuda(), gr["a/w"].cuda()
    ms, mg, mom = torch.ones(n).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    lr = torch.tensor([cfg.learning_rate]).cuda()
    for _ in range(3):
        O.rmsprop_centered_step(p, gr, slots, cfg)
        hip.rmsprop_centered_(pg, gg, ms, mg, mom, lr)
    assert_close(pg, p["a/w"], 1e-6, 1e-7, "params"); assert_close(mom, slots["a/w"]["mom"], 1e-5, 1e-9, "mom")


@pytest.mark.parametrize("kw", [dict(momentum=0.5), dict(decay=0.8, epsilon=1e-6, centered=True), dict()])
def test_rmsprop_keyword_set_of_tf_rmsprop(hip, kw):
    """air_rmsprop: every keyword of tf.train.RMSPropOptimizer (model.py:265 passes **opt_kwargs through) against the update
    written out in float64 -- TF's defaults for what is not given: decay .9, momentum 0, epsilon 1e-10, centered False."""
    full = dict(decay=0.9, momentum=0.0, epsilon=1e-10, centered=False); full.update(kw)
    gen = torch.Generator().manual_seed(9)
    n = 4099
    p, ms, mg, mom = torch.randn(n, generator=gen).double(), torch.ones(n).double(), torch.zeros(n).double(), torch.zeros(n).double()
    pg, msg, mgg, momg = p.float().cuda(), ms.float().cuda(), mg.float().cuda(), mom.float().cuda()
    lr = torch.tensor([3e-3]).cuda()
    for it in range(3):
        g = torch.randn(n, generator=gen)
        gd = g.double()
        ms = full["decay"] * ms + (1 - full["decay"]) * gd * gd
        mg = full["decay"] * mg + (1 - full["decay"]) * gd
        denom = ms - mg * mg if full["centered"] else ms
        mom = full["momentum"] * mom + 3e-3 * gd / torch.sqrt(denom + full["epsilon"])
        p = p - mom
        hip.rmsprop_centered_(pg, g.cuda(), msg, mgg, momg, lr, decay=full["decay"], momentum=full["momentum"],
                              eps=full["epsilon"], centered=full["centered"])
    assert_close(pg, p.float(), 1e-6, 1e-6, "params"); assert_close(momg, mom.float(), 1e-5, 1e-8, "mom")


@pytest.mark.parametrize("M,Hd", [(64, 256), (1024, 256)])          # 16-wave link / wide-tile link
def test_optimizer_slice_riding_on_bptt_launches(hip, M, Hd):
    """air_lstm_pointwise_bwd_opt / air_lstm_step_bwd_opt: the launch's own result is unchanged and elements [lo, hi) of the
    flat buffers -- and only those -- receive air_rmsprop_centered's update (two learning rates around n_model)."""
    gen = torch.Generator().manual_seed(M + Hd)
    rn = lambda *s_: torch.randn(*s_, generator=gen).cuda()
    n, lo, hi, n_model = 40_000, 4_000, 29_996, 20_000
    p0, g = rn(n), rn(n)
    ms0, mg0, mom0 = torch.rand(n, generator=gen).cuda() + 1.0, rn(n) * 0.1, rn(n) * 0.01
    lr = torch.tensor([1e-3]).cuda()
    ref = [t.clone() for t in (p0, ms0, mg0, mom0)]
    for a, b, mult in ((lo, n_model, 1.0), (n_model, hi, 10.0)):            # reference: the stand-alone update, segment by segment
        hip.rmsprop_centered_(ref[0][a:b], g[a:b], ref[1][a:b], ref[2][a:b], ref[3][a:b], lr, lr_mult=mult)
    act = torch.sigmoid(rn(M, 4 * Hd)); c0, c1, dh, dc = rn(M, Hd), rn(M, Hd), rn(M, Hd), rn(M, Hd)
    w_h = rn(Hd, 4 * Hd) / 16; dgn = rn(M, 4 * Hd)
    for which in ("pointwise", "link"):
        bufs = [t.clone() for t in (p0, ms0, mg0, mom0)]
        sl = hip.rmsprop_slice(bufs[0], g, bufs[1], bufs[2], bufs[3], lo, hi, n_model, lr, lr_mult_tail=10.0)
        if which == "pointwise":
            out = hip.lstm_pointwise_bwd_opt(act, c0, c1, dh, dc, sl)
            plain = hip.lstm_pointwise_bwd(act, c0, c1, dh, dc)
        else:
            out = hip.lstm_step_bwd_opt(dgn, w_h, dh, None, dc, act, c0, c1, sl)
            plain = hip.lstm_step_bwd(dgn, w_h, dh, None, dc, act, c0, c1)[:2]
        torch.cuda.synchronize()
        for o, q in zip(out, plain):
            assert torch.equal(o, q), which
        for b_, r_, o_, nm in zip(bufs, ref, (p0, ms0, mg0, mom0), ("p", "ms", "mg", "mom")):
            assert torch.equal(b_[:lo], o_[:lo]) and torch.equal(b_[hi:], o_[hi:]), (which, nm, "outside the slice")
            assert_close(b_[lo:hi], r_[lo:hi], 1e-6, 1e-7, which + " " + nm)     # (the stand-alone kernel contracts differently)
            assert not torch.equal(b_[lo:hi], o_[lo:hi])


def test_rng_statistics_and_advance(hip):
    state = torch.tensor([1234, 0], dtype=torch.int64).cuda()
    n = 1 << 20
    z = torch.empty(n).cuda(); u = torch.empty(n).cuda()
    hip.rng_fill(state, z, u)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    assert abs(u.mean().item() - 0.5) < 2e-3 and 0 <= u.min().item() and u.max().item() < 1
    assert abs((z ** 4).mean().item() - 3.0) < 0.05
    assert state.cpu()[1].item() == 2 * (n // 4)
    z2 = torch.empty(n).cuda()
    hip.rng_fill(state, z2, None)
    assert not torch.equal(z, z2)
    state2 = torch.tensor([1234, 0], dtype=torch.int64).cuda()
    z3 = torch.empty(n).cuda(); u3 = torch.empty(n).cuda()
    hip.rng_fill(state2, z3, u3)
    assert torch.equal(z, z3) and torch.equal(u, u3)                          # counter-based: reproducible


def test_small_utils(hip):
    x = torch.randn(7, 13).cuda()
    assert torch.equal(hip.tile_rows(x[0], 5), x[0][None].expand(5, -1))
    assert_close(hip.colsum(x), x.double().cpu().sum(0), 1e-6, 1e-6, "colsum")
    assert torch.equal(hip.fill_(torch.empty(100).cuda(), 2.5).cpu(), torch.full((100,), 2.5))
    assert_close(hip.axpby(x, 2.0, x, -0.5), 1.5 * x.double().cpu(), 1e-6, 1e-6, "axpby")


def test_hipgraph_capture_replay(hip):
    """A captured sequence of C-ABI launches replays with fresh inputs (static buffers)."""
    import ctypes
    from attend_infer_repeat_amd import _lib
    lib = hip.lib()
    s = torch.cuda.Stream()
    x = torch.randn(64, 400).cuda(); w = torch.randn(400, 256).cuda(); b = torch.randn(256).cuda()
    hip.workspace()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        y = hip.linear_fwd(x, w, b, 1)            # warm-up (allocates y)
        s.synchronize()
        sp = ctypes.c_void_p(s.cuda_stream)
        _lib.check(lib.air_graph_begin_capture(sp))
        wsb = ctypes.c_size_t(hip.workspace().numel() * 4)
        _lib.check(lib.air_linear_fwd(hip._p(x), hip._p(w), hip._p(b), hip._p(y), 64, 400, 256, 1,
                                      hip._p(hip.workspace()), wsb, sp))
        exe = ctypes.c_void_p()
        _lib.check(lib.air_graph_end_capture(sp, ctypes.byref(exe)))
        x.copy_(torch.randn(64, 400).cuda())
        _lib.check(lib.air_graph_launch(exe, sp))
        s.synchronize()
    ref = torch.nn.functional.elu(x.double().cpu() @ w.double().cpu() + b.double().cpu())
    assert_close(y, ref, 2e-5, 2e-4, "graph replay")
    _lib.check(lib.air_graph_destroy(exe))


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_k_split_consumer_prologue(hip, prec):
    """A long-K product split in two halves (two ordinary problems of one grouped launch) and reduced by its CONSUMER:
    a = elu(slab0 + slab1 + bias) formed in the A-operand loader of the next product, which also stores the reduced activation;
    a second consumer picks a partial product up with beta = 1.  Checked against the unsplit computation in float64."""
    torch.manual_seed(7)
    M, K, E, N = 64, 2500, 256, 320
    x, w0, b0 = torch.rand(M, K).cuda(), (torch.randn(K, E) / 50).cuda(), (0.1 * torch.randn(E)).cuda()
    w1, b1 = (torch.randn(E, N) / 16).cuda(), (0.1 * torch.randn(N)).cuda()
    kh = (K // 2) // 16 * 16
    (s0, _), (s1, _) = hip.gemm_grouped([dict(A=x[:, :kh], B=w0[:kh]), dict(A=x[:, kh:], B=w0[kh:])], precision=prec)
    act = torch.empty(M, E, device="cuda")
    (y, _), = hip.gemm_grouped([dict(A=s0, B=w1, bias=b1, epilogue=hip.EPI_BIAS_ELU, A2=s1, a_bias=b0, a_elu=True, a_out=act)],
                               precision=prec)
    r = lambda t: t.double() if prec == 0 else t.bfloat16().double()          # operand rounding of the bf16 mode
    h_ref = torch.nn.functional.elu(r(x) @ r(w0) + b0.double())
    y_ref = torch.nn.functional.elu(r(h_ref.float()) @ r(w1) + b1.double())
    tol = 2e-5 if prec == 0 else 3e-3
    assert ((act.double() - h_ref).abs().max() / h_ref.abs().max()).item() < tol
    assert ((y.double() - y_ref).abs().max() / y_ref.abs().max()).item() < tol
    # beta = 1 pick-up of a partial product written into C beforehand
    c = s1.clone()
    hip.gemm_grouped([dict(A=x[:, :kh], B=w0[:kh], out=c, beta=1.0, bias=b0, epilogue=hip.EPI_BIAS)], precision=prec)
    full = r(x) @ r(w0) + b0.double()
    assert ((c.double() - full).abs().max() / full.abs().max()).item() < tol


@pytest.mark.parametrize("mirrors", ["ab", "a", "b", "none"])
@pytest.mark.parametrize("layout", ["NN", "NT", "TN"])
def test_gemm_bf16_data_path(hip, layout, mirrors):
    """BASELINE configs[4], the bf16 DATA path: operands read from bf16 mirrors (A16 / B16: any combination -- an operand
    without a mirror is fetched as fp32 and rounded in registers), products on v_mfma_f32_16x16x32_bf16, the epilogue writing
    the bf16 mirror of C.  Against float64 products of the bf16-rounded operands: K with 32-, 8- and sub-8 tails (2500, 400,
    50, 100, 12), ragged M / N, every epilogue, beta, bias gradients (summed from the UNROUNDED fp32 gradient), single, grouped
    and whole-step (more than 8 problems, per-problem operand kinds) launches; the mirror of C must be exactly bf16(C)."""
    gen = torch.Generator().manual_seed(31 + len(layout) * ord(layout[1]) + len(mirrors))
    rn = lambda *s_: torch.randn(*s_, generator=gen).cuda()
    r = lambda t: t.cpu().to(torch.bfloat16).double()
    h16 = lambda t: t.to(torch.bfloat16)
    ua, ub = "a" in mirrors, "b" in mirrors
    atol = 2e-3

    def prob(A, B, **kw):
        d = dict(A=A, B=B, **kw)
        if ua:
            d["A16"] = h16(A)
        if ub:
            d["B16"] = h16(B)
        return d

    def check_c16(out, c16, what):
        assert torch.equal(c16, out.to(torch.bfloat16)), what

    if layout == "NN":
        x = rn(3000, 256); w = rn(256, 400) / 16; b = rn(400); aux = rn(3000, 400)
        x2 = rn(1024, 2500); w2 = rn(2500, 256) / 50; c0 = rn(1024, 256)
        x3 = rn(2048, 12); w3 = rn(12, 100)
        c16 = torch.zeros(3000, 400, dtype=torch.bfloat16, device="cuda")
        outs = hip.gemm_grouped([prob(x, w, bias=b, epilogue=hip.EPI_BIAS_ELU, C16=c16)], precision=1)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x) @ r(w) + b.cpu().double()), 1e-5, atol, "single bias+elu")
        check_c16(outs[0][0], c16, "mirror of C, single")
        c16b = torch.zeros(1024, 256, dtype=torch.bfloat16, device="cuda")
        outs = hip.gemm_grouped([prob(x, w, bias=b, aux=aux, epilogue=hip.EPI_ADD_AUX_ELU),
                                 prob(x2, w2, beta=1.0, out=c0.clone(), C16=c16b), prob(x3, w3)], precision=1)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x) @ r(w) + b.cpu().double() + aux.cpu().double()), 1e-5, atol, "add aux elu")
        assert_close(outs[1][0], r(x2) @ r(w2) + c0.cpu().double(), 1e-5, atol, "K = 2500 (tail of 4) + beta")
        check_c16(outs[1][0], c16b, "mirror of C, grouped")
        assert_close(outs[2][0], r(x3) @ r(w3), 1e-5, atol, "K = 12")
        x4 = rn(3072, 50); w4 = rn(50, 256) / 7; b4 = rn(256)
        x5 = rn(3072, 100); w5 = rn(100, 256) / 10
        outs = hip.gemm_grouped([prob(x4, w4, bias=b4, epilogue=hip.EPI_BIAS_ELU), prob(x5, w5)], precision=1)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x4) @ r(w4) + b4.cpu().double()), 1e-5, atol, "K = 50")
        assert_close(outs[1][0], r(x5) @ r(w5), 1e-5, atol, "K = 100")
    elif layout == "NT":
        g = rn(3000, 400); w = rn(1024, 400) / 16; y = rn(3000, 1024)
        g2 = rn(1024, 1024); w2 = rn(256, 1024) / 32; c0 = rn(1024, 256)
        c16 = torch.zeros(1024, 256, dtype=torch.bfloat16, device="cuda")
        outs = hip.gemm_grouped([prob(g2, w2, tb=True, beta=1.0, out=c0.clone(), C16=c16)], precision=1)
        assert_close(outs[0][0], r(g2) @ r(w2).t() + c0.cpu().double(), 1e-5, atol, "single beta")
        check_c16(outs[0][0], c16, "mirror of C")
        c16b = torch.zeros(3000, 1024, dtype=torch.bfloat16, device="cuda")
        outs = hip.gemm_grouped([prob(g, w, tb=True, epilogue=hip.EPI_MUL_DELU, aux=y, C16=c16b), prob(g2, w2, tb=True)], precision=1)
        d = torch.where(y.cpu() > 0, torch.ones_like(y.cpu()), y.cpu() + 1).double()
        assert_close(outs[0][0], (r(g) @ r(w).t()) * d, 1e-5, atol, "mul delu, K = 400 (tail of 16)")
        check_c16(outs[0][0], c16b, "mirror of C, grouped")
        assert_close(outs[1][0], r(g2) @ r(w2).t(), 1e-5, atol, "plain")
        # short K on the 32x64 tiles (dX of a 50-wide layer) and a 100-deep one
        g3 = rn(3072, 256); w3 = rn(50, 256) / 16
        g4 = rn(3072, 100); w4 = rn(256, 100) / 10
        outs = hip.gemm_grouped([prob(g3, w3, tb=True)], precision=1)
        assert_close(outs[0][0], r(g3) @ r(w3).t(), 1e-5, atol, "N = 50")
        outs = hip.gemm_grouped([prob(g4, w4, tb=True)], precision=1)
        assert_close(outs[0][0], r(g4) @ r(w4).t(), 1e-5, atol, "K = 100")
    else:
        x = rn(3072, 2500); g = rn(3072, 256); x2 = rn(1024, 260); g2 = rn(1024, 1024)
        outs = hip.gemm_grouped([prob(x, g, ta=True, colsum=True)], precision=1)
        assert_close(outs[0][0], r(x).t() @ r(g), 1e-5, atol * 10, "single dW")
        assert_close(outs[0][1], g.cpu().double().sum(0), 1e-5, 1e-3, "single db (from the fp32 gradient)")
        outs = hip.gemm_grouped([prob(x, g, ta=True, colsum=True), prob(x2, g2, ta=True, colsum=True)], precision=1)
        assert_close(outs[0][0], r(x).t() @ r(g), 1e-5, atol * 10, "dW")
        assert_close(outs[1][0], r(x2).t() @ r(g2), 1e-5, atol * 10, "dW ragged M")
        assert_close(outs[1][1], g2.cpu().double().sum(0), 1e-5, 1e-3, "db")
        # more than 8 problems: per-problem operand kinds (every other problem has no A mirror, every third no B mirror)
        xs = [rn(1024 if i % 2 else 3072, 64 * (1 + i % 3)) for i in range(11)]
        gs = [rn(x_.shape[0], 128 if i % 4 else 256) for i, x_ in enumerate(xs)]
        order = sorted(range(11), key=lambda i: -xs[i].shape[0])
        probs = []
        for i in order:
            d = dict(A=xs[i], B=gs[i], ta=True, colsum=(i % 2 == 0))
            if ua and i % 2 == 0:
                d["A16"] = h16(xs[i])
            if ub and i % 3 != 0:
                d["B16"] = h16(gs[i])
            probs.append(d)
        outs = hip.gemm_grouped(probs, precision=1)
        for o, i in zip(outs, order):
            assert_close(o[0], r(xs[i]).t() @ r(gs[i]), 1e-5, atol * 10, "big group dW %d" % i)
            if i % 2 == 0:
                assert_close(o[1], gs[i].cpu().double().sum(0), 1e-5, 1e-3, "big group db %d" % i)


def test_step_epilogue_shadow_and_f32_to_bf16(hip):
    """bf16 shadow of the parameters: air_step_epilogue_shadow == air_step_epilogue on the fp32 state and additionally leaves
    bf16(p) in the shadow; air_f32_to_bf16 rounds like torch (RNE), vector and scalar paths."""
    import ctypes
    from attend_infer_repeat_amd import hip as H
    L = H.lib()
    gen = torch.Generator().manual_seed(2)
    n, n_model = 40_004, 30_000
    mk = lambda: [torch.randn(n, generator=gen).cuda(), torch.randn(n, generator=gen).cuda(), (torch.rand(n, generator=gen) + 1).cuda(),
                  (torch.randn(n, generator=gen) * 0.1).cuda(), (torch.randn(n, generator=gen) * 0.01).cuda()]
    a = mk(); b = [t.clone() for t in a]
    lr = torch.tensor([1e-3]).cuda()
    step = torch.zeros(1, dtype=torch.int64).cuda(); rng = torch.zeros(2, dtype=torch.int64).cuda()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = H._p
    common = lambda t: (P(t[0]), P(t[1]), P(t[2]), P(t[3]), P(t[4]), ctypes.c_size_t(n_model), ctypes.c_size_t(n), P(lr), 10.0, 0.9, 0.9,
                        1e-10, 1.0)
    assert L.air_step_epilogue(*common(a), P(step), P(rng), ctypes.c_uint64(7), sp) == 0
    shadow = torch.zeros(n, dtype=torch.bfloat16).cuda()
    assert L.air_step_epilogue_shadow(*common(b), None, None, ctypes.c_uint64(0), ctypes.c_void_p(shadow.data_ptr()), sp) == 0
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(shadow, b[0].to(torch.bfloat16))
    assert step.item() == 1 and rng[1].item() == 7
    for m in (4096, 4099):
        src = torch.randn(m + 1, generator=gen).cuda()[1:] if m == 4099 else torch.randn(m, generator=gen).cuda()
        dst = torch.zeros(m, dtype=torch.bfloat16).cuda()
        assert L.air_f32_to_bf16(P(src), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(m), sp) == 0
        torch.cuda.synchronize()
        assert torch.equal(dst, src.to(torch.bfloat16))


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("layout", ["NN", "NT", "TN"])
def test_gemm_wide_tiles(hip, precision, layout):
    """Throughput-regime groups with ONE operand layout take the wide-tile kernels (interleaved 16-byte loads, gemm_wide_body):
    single and grouped launches, every epilogue, beta, colsum, ragged M / N (N % 64 != 0, M % 16 != 0), K with a 4-deep tail
    (2500) and K < 16, row views with a leading dimension."""
    gen = torch.Generator().manual_seed(11 + precision + len(layout) * ord(layout[0]))
    r = (lambda t: t.cpu().to(torch.bfloat16).double()) if precision else (lambda t: t.cpu().double())
    rn = lambda *s: torch.randn(*s, generator=gen).cuda()
    atol = 2e-3
    if layout == "NN":
        x = rn(3000, 256); w = rn(256, 400) / 16; b = rn(400); aux = rn(3000, 400)
        x2big = rn(1024, 2500 + 12); x2 = x2big[:, 8:8 + 2500]; w2 = rn(2500, 256) / 50; c0 = rn(1024, 256)
        x3 = rn(2048, 12); w3 = rn(12, 100)
        outs = hip.gemm_grouped([dict(A=x, B=w, bias=b, epilogue=hip.EPI_BIAS_ELU)], precision=precision)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x) @ r(w) + b.cpu().double()), 1e-5, atol, "single bias+elu")
        outs = hip.gemm_grouped([
            dict(A=x, B=w, bias=b, aux=aux, epilogue=hip.EPI_ADD_AUX_ELU),
            dict(A=x2, B=w2, beta=1.0, out=c0.clone()),
            dict(A=x3, B=w3),
        ], precision=precision)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x) @ r(w) + b.cpu().double() + aux.cpu().double()), 1e-5, atol, "add aux elu")
        assert_close(outs[1][0], r(x2) @ r(w2) + c0.cpu().double(), 1e-5, atol, "K tail + beta + view")
        assert_close(outs[2][0], r(x3) @ r(w3), 1e-5, atol, "K < 16")
        # K % 4 != 0 and leading dimensions that are not multiples of 4 (16-byte loads from 4-byte aligned addresses; the lane
        # group that straddles the end of K loads element by element): the decoder's first layer (K = 50) and the baseline's
        # latent columns (K = 677), the second through a row view that starts one float into its buffer
        x4 = rn(3072, 50); w4 = rn(50, 256) / 7; b4 = rn(256)
        x5big = rn(1024, 679); x5 = x5big[:, 1:678]; w5 = rn(677, 256) / 26
        outs = hip.gemm_grouped([dict(A=x4, B=w4, bias=b4, epilogue=hip.EPI_BIAS_ELU), dict(A=x5, B=w5)], precision=precision)
        assert_close(outs[0][0], torch.nn.functional.elu(r(x4) @ r(w4) + b4.cpu().double()), 1e-5, atol, "K = 50")
        assert_close(outs[1][0], r(x5) @ r(w5), 1e-5, atol, "K = 677, unaligned rows")
    elif layout == "NT":
        g = rn(3000, 400); w = rn(1024, 400) / 16; y = rn(3000, 1024)
        g2 = rn(1024, 1024); w2 = rn(256, 1024) / 32; c0 = rn(1024, 256)
        outs = hip.gemm_grouped([dict(A=g2, B=w2, tb=True, beta=1.0, out=c0.clone())], precision=precision)
        assert_close(outs[0][0], r(g2) @ r(w2).t() + c0.cpu().double(), 1e-5, atol, "single beta")
        outs = hip.gemm_grouped([dict(A=g, B=w, tb=True, epilogue=hip.EPI_MUL_DELU, aux=y),
                                 dict(A=g2, B=w2, tb=True)], precision=precision)
        d = torch.where(y.cpu() > 0, torch.ones_like(y.cpu()), y.cpu() + 1).double()
        assert_close(outs[0][0], (r(g) @ r(w).t()) * d, 1e-5, atol, "mul delu")
        assert_close(outs[1][0], r(g2) @ r(w2).t(), 1e-5, atol, "plain")
    else:
        x = rn(3072, 2500); g = rn(3072, 256); x2 = rn(1024, 260); g2 = rn(1024, 1024)
        outs = hip.gemm_grouped([dict(A=x, B=g, ta=True, colsum=True)], precision=precision)
        assert_close(outs[0][0], r(x).t() @ r(g), 1e-5, atol * 10, "single dW")
        assert_close(outs[0][1], g.cpu().double().sum(0), 1e-5, 1e-3, "single db")
        outs = hip.gemm_grouped([dict(A=x, B=g, ta=True, colsum=True), dict(A=x2, B=g2, ta=True, colsum=True)], precision=precision)
        assert_close(outs[0][0], r(x).t() @ r(g), 1e-5, atol * 10, "dW")
        assert_close(outs[1][0], r(x2).t() @ r(g2), 1e-5, atol * 10, "dW ragged M")
        assert_close(outs[1][1], g2.cpu().double().sum(0), 1e-5, 1e-3, "db")
        # more than 8 problems: the whole-step form (dynamic descriptor index), long-K problems first
        xs = [rn(1024 if i % 2 else 3072, 64 * (1 + i % 3)) for i in range(11)]
        gs = [rn(x_.shape[0], 128 if i % 4 else 256) for i, x_ in enumerate(xs)]
        order = sorted(range(11), key=lambda i: -xs[i].shape[0])
        outs = hip.gemm_grouped([dict(A=xs[i], B=gs[i], ta=True, colsum=(i % 2 == 0)) for i in order], precision=precision)
        for o, i in zip(outs, order):
            assert_close(o[0], r(xs[i]).t() @ r(gs[i]), 1e-5, atol * 10, "big group dW %d" % i)
            if i % 2 == 0:
                assert_close(o[1], gs[i].cpu().double().sum(0), 1e-5, 1e-3, "big group db %d" % i)
        # rows of A that are not 16-byte aligned (leading dimension 50): the first 48 of the 50 weight-gradient rows
        x3 = rn(3072, 50); g3 = rn(3072, 256)
        outs = hip.gemm_grouped([dict(A=x3[:, :48], B=g3, ta=True, colsum=True), dict(A=x2, B=g2, ta=True)], precision=precision)
        assert_close(outs[0][0], r(x3[:, :48]).t() @ r(g3), 1e-5, atol * 10, "dW, lda = 50")
        assert_close(outs[0][1], g3.cpu().double().sum(0), 1e-5, 1e-3, "db")


@pytest.mark.parametrize("mirror_in", [False, True])
def test_lstm_steps_on_the_bf16_data_path(hip, mirror_in):
    """air_lstm_step_fwd_bf16 / air_lstm_step_bwd_bf16 / air_lstm_pointwise_bwd_bf16 (W_h from the bf16 shadow, h_prev / dgates_next
    from mirrors or from fp32, v_mfma_f32_16x16x32_bf16) against the entries that fetch fp32 and round in registers
    (precision = bf16): the operands a product sees are identical, so the results agree to summation order, and every mirror the
    kernels write is exactly bf16 of the fp32 output next to it."""
    import ctypes
    from attend_infer_repeat_amd import hip as H
    L, P = H.lib(), H._p
    gen = torch.Generator().manual_seed(17)
    rn = lambda *s_: torch.randn(*s_, generator=gen).cuda()
    M, Hd = 1024, 256
    h_prev, c_prev, gx = rn(M, Hd) * 0.5, rn(M, Hd), rn(M, 4 * Hd)
    w_h = rn(Hd, 4 * Hd) / 16
    w16, h16_in = w_h.to(torch.bfloat16), h_prev.to(torch.bfloat16)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    v = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    h_ref, c_ref, act_ref = hip.lstm_step_fwd(h_prev, c_prev, w_h, gx, precision=1)
    h, c, act = torch.empty_like(c_prev), torch.empty_like(c_prev), torch.empty_like(gx)
    h16 = torch.zeros(M, Hd, dtype=torch.bfloat16, device="cuda")
    st = L.air_lstm_step_fwd_bf16(P(h_prev), v(h16_in if mirror_in else None), P(c_prev), v(w16), 4 * Hd, P(gx), 4 * Hd, P(h), v(h16),
                                  P(c), P(act), M, Hd, 1.0, sp)
    assert st == 0
    torch.cuda.synchronize()
    assert_close(h, h_ref, 1e-4, 1e-5, "h"); assert_close(c, c_ref, 1e-4, 1e-5, "c"); assert_close(act, act_ref, 1e-4, 1e-5, "gate_act")
    assert torch.equal(h16, h.to(torch.bfloat16))
    # one BPTT link
    dg_next, dh_a, dh_b, dc_in, dgx_in = rn(M, 4 * Hd), rn(M, Hd), rn(M, Hd), rn(M, Hd), rn(M, 4 * Hd)
    dg_ref, dcp_ref, dgx_ref = hip.lstm_step_bwd(dg_next, w_h, dh_a, dh_b, dc_in, act_ref, c_prev, c_ref, dgx_in=dgx_in, want_dgx=True,
                                                 precision=1)
    dg, dcp, dgx = torch.empty_like(gx), torch.empty_like(c_prev), torch.empty_like(gx)
    dg16 = torch.zeros(M, 4 * Hd, dtype=torch.bfloat16, device="cuda"); dgx16 = torch.zeros_like(dg16)
    dgn16 = dg_next.to(torch.bfloat16)
    st = L.air_lstm_step_bwd_bf16(P(dg_next), v(dgn16 if mirror_in else None), v(w16), P(dh_a), P(dh_b), P(dc_in), P(act_ref), P(c_prev),
                                  P(c_ref), P(dgx_in), P(dg), v(dg16), P(dcp), P(dgx), v(dgx16), M, Hd, sp)
    assert st == 0
    torch.cuda.synchronize()
    scale = dg_ref.abs().max().item()
    assert_close(dg, dg_ref, 1e-4, 1e-5 * scale, "dgates"); assert_close(dcp, dcp_ref, 1e-4, 1e-5 * scale, "dc_prev")
    assert_close(dgx, dgx_ref, 1e-4, 1e-5 * scale, "dgx")
    assert torch.equal(dg16, dg.to(torch.bfloat16)) and torch.equal(dgx16, dgx.to(torch.bfloat16))
    # the pointwise backward of the last step with its mirror
    pw_ref = torch.empty_like(gx); pw_dc = torch.empty_like(c_prev)
    assert L.air_lstm_pointwise_bwd(P(act_ref), P(c_prev), P(c_ref), P(dh_a), P(dh_b), None, P(pw_ref), P(pw_dc), M, Hd, sp) == 0
    pw, pw_dc2, pw16 = torch.empty_like(gx), torch.empty_like(c_prev), torch.zeros(M, 4 * Hd, dtype=torch.bfloat16, device="cuda")
    assert L.air_lstm_pointwise_bwd_bf16(P(act_ref), P(c_prev), P(c_ref), P(dh_a), P(dh_b), None, P(pw), v(pw16), P(pw_dc2), M, Hd, sp) == 0
    torch.cuda.synchronize()
    assert torch.equal(pw, pw_ref) and torch.equal(pw_dc, pw_dc2) and torch.equal(pw16, pw.to(torch.bfloat16))


@pytest.mark.parametrize("widths,rows", [((400, 256, 256, 50), 3072), ((1, 128, 256), 1024), ((100, 256, 256, 400), 100), ((64, 128, 256), 37),
                                         ((8, 24, 40), 16), ((36, 44, 100, 52), 33)])
def test_mlp_dx_chain_matches_per_layer_products(hip, widths, rows):
    """Round 6: air_mlp_dx_chain_bf16 -- a whole dX chain dA_{l-1} = (dA_l . W_l^T) * elu'(out_{l-1}) per slab of 16 rows in ONE launch --
    against the same chain evaluated layer by layer in float64 on the bf16-rounded operands (what the per-layer launches of the bf16 data
    path compute: every layer reads bf16 of the previous fp32 result): the configs[4] decoder chain, the baseline's (1-wide output layer:
    the scalar path), the glimpse encoder's, ragged row counts and widths that are not multiples of 32; two chains in one launch."""
    import ctypes
    from attend_infer_repeat_amd import _lib
    lib = hip.lib()
    rng = np.random.default_rng(len(widths) * 1000 + rows)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)

    def make(widths, rows):
        g_in = torch.tensor(rng.standard_normal((rows, widths[0])).astype(np.float32), device="cuda")
        layers, keep = [], [g_in]
        cur = bf(g_in).double()
        for l in range(len(widths) - 1):
            n_in, n_out = widths[l], widths[l + 1]
            w = torch.tensor((rng.standard_normal((n_out, n_in)) / np.sqrt(n_in)).astype(np.float32), device="cuda")
            w16 = w.to(torch.bfloat16).contiguous()
            aux = torch.tensor(rng.standard_normal((rows, n_out)).astype(np.float32), device="cuda") if l < len(widths) - 2 else None
            out = torch.full((rows, n_out), float("nan"), device="cuda"); out16 = torch.zeros(rows, n_out, dtype=torch.bfloat16, device="cuda")
            ref = cur @ w16.float().double().t()
            if aux is not None:
                ref = ref * torch.where(aux > 0, torch.ones_like(aux), aux + 1.0).double()
            cur = bf(ref.float()).double()
            layers.append((w16, aux, out, out16, n_in, n_out, ref))
            keep += [w, w16, aux, out, out16]
        return g_in, layers, keep

    sets = [make(widths, rows), make((64, 256, 96), 50)]                      # a second, short chain in the same launch
    arr = (_lib.AirDxChain * len(sets))()
    for ci, (g_in, layers, _) in enumerate(sets):
        arr[ci].g_in, arr[ci].ld_in, arr[ci].rows, arr[ci].n_layers = g_in.data_ptr(), g_in.shape[1], g_in.shape[0], len(layers)
        for li, (w16, aux, out, out16, n_in, n_out, _) in enumerate(layers):
            y = arr[ci].layer[li]
            y.w_bf16, y.aux, y.out, y.out_bf16 = w16.data_ptr(), (aux.data_ptr() if aux is not None else None), out.data_ptr(), out16.data_ptr()
            y.n_in, y.n_out, y.ldaux, y.ldout = n_in, n_out, n_out, n_out
            assert lib.air_mlp_dx_chain_fits(n_in, n_out) == 1
    _lib.check(lib.air_mlp_dx_chain_bf16(arr, len(sets), hip._stream()), "air_mlp_dx_chain_bf16")
    torch.cuda.synchronize()
    for g_in, layers, _ in sets:
        for w16, aux, out, out16, n_in, n_out, ref in layers:
            scale = ref.abs().max().item() + 1e-30
            assert torch.isfinite(out).all()
            # (a value that sits on a bf16 rounding boundary may round the other way in the next layer's input: bf16 ulp = 2^-8 relative)
            err = ((out.double() - ref).abs().max() / scale).item()
            assert err < 2e-2 if (layers[0][0] is not w16) else err < 1e-5, (n_in, n_out, err)
            assert torch.equal(out16, out.to(torch.bfloat16))
    assert lib.air_mlp_dx_chain_fits(38, 64) == 0 and lib.air_mlp_dx_chain_fits(2048, 64) == 0


@pytest.mark.parametrize("case", range(24))
def test_canvas_kernels_random_shapes_and_transforms(hip, case):
    """Seeded sweep over canvas / glimpse shapes (odd sizes, w % 4 in {0..3}, glimpses larger than the canvas), step counts, batch sizes
    (grid-strided from ~600 units) and transforms (mirrored, tiny, huge, far off-canvas): the unrolled forward is the oracle's sequential
    accumulation BIT FOR BIT (per-step canvases and the final one), the fused forward + backward launch equals the two launches, and the
    recompute-form backward equals the stored-canvas one -- all bitwise; the backward itself against float64 autograd of the oracle."""
    rng = np.random.default_rng(1000 + case)
    H, W = int(rng.integers(3, 41)), int(rng.integers(3, 41))
    h, w = int(rng.integers(2, 18)), int(rng.integers(2, 18))
    T = int(rng.integers(1, 6))
    B = int(rng.choice([1, 2, 5, 17, 64, 230]))
    glm = rng.standard_normal((T, B, h, w)).astype(np.float32)
    where = rand_where(T * B, rng, wide=True).reshape(T, B, 4)
    k = rng.integers(0, T * B, 6)
    flat = where.reshape(-1, 4)
    flat[k[0]] = [1e-3, 0.0, 1e-3, 0.0]            # the whole glimpse inside one canvas pixel
    flat[k[1]] = [25.0, 0.3, 40.0, -0.2]           # one glimpse pixel covers the canvas
    flat[k[2]] = [0.5, 3.0, 0.5, -3.0]             # far off the canvas: contributes exactly zero
    flat[k[3]] = [-1.0, 0.0, -1.0, 0.0]            # mirrored on both axes
    pres = np.cumprod(rng.integers(0, 2, (T, B)), 0).astype(np.float32); pres[:, 0] = 1.0
    obs = rng.random((B, H, W)).astype(np.float32)
    mult, std = 0.5, 0.3
    canvas = np.zeros((B, H, W), np.float32)
    steps = []
    for t in range(T):
        canvas = canvas + pres[t][:, None, None] * C.st_write_fwd(glm[t], where[t], (H, W))
        steps.append(canvas.copy())
    st, final, rec = hip.canvas_unroll_fwd(g(glm), g(where), g(pres), (H, W), obs=g(obs), mult=mult, std=std)
    np.testing.assert_array_equal(st.cpu().numpy(), np.stack(steps))
    np.testing.assert_array_equal(final.cpu().numpy(), steps[-1])
    dg, dwhere = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), final, mult, std, 1.0 / B)
    dg2, dwhere2 = hip.canvas_unroll_bwd(g(glm), g(where), g(pres), g(obs), None, mult, std, 1.0 / B)
    assert torch.equal(dg, dg2) and torch.equal(dwhere, dwhere2)
    if B * T <= 4096:                                  # the one-launch form of the latency regime
        lib, p = hip.lib(), hip._p
        from attend_infer_repeat_amd import _lib
        nb = int(lib.air_canvas_unroll_bands(B, H))
        d_glm, d_where, d_pres, d_obs = g(glm), g(where), g(pres), g(obs)      # (held: a raw pointer does not keep a tensor alive)
        for ns in (1, 2, 3):          # workgroups per backward unit: disjoint dglimpse rows, dwhere as `ns` slabs whose sum is the gradient
            if lib.air_canvas_unroll_fwd_bwd_fits(nb, ns, T, B, H, W, h, w) != 1:
                continue
            st3 = torch.empty(T, B, H, W, device="cuda"); fin3 = torch.empty(B, H, W, device="cuda")
            parts = torch.empty(nb, B, device="cuda"); dg3 = torch.empty_like(dg); dw3 = torch.full((ns,) + tuple(dwhere.shape), float("nan"), device="cuda")
            _lib.check(lib.air_canvas_unroll_fwd_bwd(p(d_glm), p(d_where), p(d_pres), p(d_obs), p(st3), p(fin3), p(parts), nb, p(dg3), p(dw3),
                                                     ns, T, B, H, W, h, w, mult, std, 1.0 / B, hip._stream()), "air_canvas_unroll_fwd_bwd")
            torch.cuda.synchronize()
            assert torch.equal(st3, st) and torch.equal(fin3, final) and torch.equal(dg3, dg)
            if ns == 1:
                assert torch.equal(dw3[0], dwhere)
            else:
                assert_close(dw3.sum(0), dwhere, 2e-4, 1e-5 * float(dwhere.abs().max()) + 1e-7, "dwhere from %d slabs" % ns)
            assert_close(parts.sum(0), rec, 1e-5, 1e-3, "rec from band shares")
    tg = torch.tensor(glm, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(where, dtype=torch.float64, requires_grad=True)
    cv = sum(torch.tensor(pres[t], dtype=torch.float64)[:, None, None] * O.st_write(tg[t], tw[t], (H, W)) for t in range(T))
    nll = 0.5 * ((torch.tensor(obs, dtype=torch.float64) - mult * cv) / std) ** 2 + 0.5 * np.log(2 * np.pi) + np.log(std)
    gg, gw = torch.autograd.grad(nll.sum((1, 2)).mean(), [tg, tw])
    # (the four extreme transforms stay out of the float64 comparison: with a scale of 1e-3 or 40 the fp32 and fp64 sampling
    #  coordinates fall into different source cells, and d/d(scale) carries a 1/s^2 -- they are covered by the bitwise checks above)
    keep = np.ones(T * B, bool); keep[k[:4]] = False
    keep = keep.reshape(T, B)
    assert_close(dg.cpu().numpy()[keep], gg.numpy()[keep], 3e-4, 2e-4 * max(gg.abs().max().item(), 1e-6), "dglimpse")
    scale = gw.abs().amax(-1, keepdim=True).numpy() + 1.0
    assert_close((dwhere.cpu().numpy() / scale)[keep], (gw.numpy() / scale)[keep], 1e-3, 1e-4, "dwhere")


@pytest.mark.parametrize("case", range(16))
def test_st_read_random_shapes_and_transforms(hip, case):
    """Seeded sweep of the glimpse read over image / glimpse shapes, T glimpses per image and transforms (mirrored, tiny, huge, off
    the image): forward bit for bit the oracle's; backward against the oracle's float64 closed form (extreme transforms left to the
    forward check: fp32 and fp64 coordinates fall into different cells there)."""
    rng = np.random.default_rng(2000 + case)
    H, W = int(rng.integers(2, 61)), int(rng.integers(2, 61))
    h, w = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    B, T = int(rng.choice([1, 3, 8, 65, 300])), int(rng.integers(1, 5))
    img = rng.random((B, H, W)).astype(np.float32)
    where = rand_where(T * B, rng, wide=True)
    k = rng.integers(0, T * B, 4)
    where[k[0]] = [1e-3, 0.2, 1e-3, -0.1]
    where[k[1]] = [30.0, 0.0, 30.0, 0.0]
    where[k[2]] = [0.4, 4.0, 0.4, 4.0]
    where[k[3]] = [-1.0, 0.0, -1.0, 0.0]
    ref = C.st_read_fwd(np.tile(img, (T, 1, 1)), where, (h, w))
    out = hip.st_read_fwd(g(img), g(where), (h, w)).cpu().numpy()
    np.testing.assert_array_equal(out, ref)
    if T == 1:
        dout = rng.standard_normal((B, h, w)).astype(np.float32)
        dwhere64, dimg64 = C.st_read_bwd(img.astype(np.float64), where.astype(np.float64), dout.astype(np.float64))
        dwhere, dimg = hip.st_read_bwd(g(img), g(where), g(dout), want_dimg=True)
        keep = np.ones(B, bool); keep[k] = False
        scale = np.abs(dout).sum((1, 2))[:, None] * max(H, W) / 2 * 0.05 + 1.0
        assert_close((dwhere.cpu().numpy() / scale)[keep], (dwhere64 / scale)[keep], 2e-4, 5e-5, "dwhere")
        assert_close(dimg.cpu().numpy()[keep], dimg64[keep], 2e-4, 2e-5, "dimg")


@pytest.mark.parametrize("case", range(20))
def test_gemm_grouped_random_groups(hip, case):
    """Seeded sweep of air_gemm_grouped: 1-8 problems of random shape (1 ... ~1500 rows / columns, K from 1 to ~1100), layout, epilogue,
    beta, leading dimensions (row views, unaligned starts) and bias-gradient column sums, from a handful of tiles up to thousands (the
    wide-tile dispatch), fp32 and bf16 operands -- every result against float64 (bf16: against the rounded operands)."""
    rng = np.random.default_rng(4000 + case)
    gen = torch.Generator().manual_seed(4000 + case)
    precision = int(case % 4 == 3)
    r = (lambda t: t.to(torch.bfloat16).double()) if precision else (lambda t: t.double())
    big = case % 5 == 4                                       # a group far into the throughput regime
    n = int(rng.integers(1, 9))
    problems, refs = [], []
    for _ in range(n):
        hi = 1500 if big else 200
        M, N = int(rng.integers(1, hi)), int(rng.integers(1, hi))
        K = int(rng.choice([1, 3, 16, 50, 64, 100, 256, 400, 677, 1100]))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if rng.integers(0, 3) == 0:                           # multiples of 4 / 16: the aligned fast paths
            M, N = max(16, M // 16 * 16), max(16, N // 16 * 16)

        def operand(rows, cols):
            pad, off = int(rng.integers(0, 7)), int(rng.integers(0, 4))
            base = torch.randn(rows, cols + pad + off, generator=gen).cuda()
            return base[:, off:off + cols] if (pad or off) else base
        A = operand(K, M) if ta else operand(M, K)
        Bm = operand(N, K) if tb else operand(K, N)
        epi = int(rng.choice([hip.EPI_NONE, hip.EPI_BIAS, hip.EPI_BIAS_ELU, hip.EPI_MUL_DELU, hip.EPI_ADD_AUX, hip.EPI_ADD_AUX_ELU]))
        bias = torch.randn(N, generator=gen).cuda() if epi in (hip.EPI_BIAS, hip.EPI_BIAS_ELU, hip.EPI_ADD_AUX, hip.EPI_ADD_AUX_ELU) else None
        aux = torch.randn(M, N, generator=gen).cuda() if epi >= hip.EPI_MUL_DELU else None
        beta = float(rng.choice([0.0, 0.0, 1.0]))
        out = torch.randn(M, N, generator=gen).cuda() if beta else None
        colsum = bool(ta and not tb and rng.integers(0, 2))   # the bias gradient rides on weight-gradient products
        pr = dict(A=A, B=Bm, ta=ta, tb=tb, epilogue=epi, beta=beta, colsum=colsum)
        if bias is not None: pr["bias"] = bias
        if aux is not None: pr["aux"] = aux
        ref = (r(A.cpu()).t() if ta else r(A.cpu())) @ (r(Bm.cpu()).t() if tb else r(Bm.cpu()))
        if beta:
            ref = ref + beta * out.cpu().double(); pr["out"] = out
        if epi == hip.EPI_BIAS: ref = ref + bias.cpu().double()
        elif epi == hip.EPI_BIAS_ELU: ref = torch.nn.functional.elu(ref + bias.cpu().double())
        elif epi == hip.EPI_MUL_DELU:
            y = aux.cpu().double(); ref = ref * torch.where(y > 0, torch.ones_like(y), y + 1)
        elif epi == hip.EPI_ADD_AUX: ref = ref + aux.cpu().double() + bias.cpu().double()
        elif epi == hip.EPI_ADD_AUX_ELU: ref = torch.nn.functional.elu(ref + aux.cpu().double() + bias.cpu().double())
        problems.append(pr); refs.append((ref, Bm.cpu().double().sum(1 if tb else 0) if colsum else None, K))
    outs = hip.gemm_grouped(problems, precision=precision)
    for i, ((C_, cs), (ref, cref, K)) in enumerate(zip(outs, refs)):
        atol = 2e-5 * np.sqrt(K) * 4 + 1e-5
        assert_close(C_, ref, 2e-5, atol, f"problem {i} of {n} (K={K})")
        if cref is not None:
            assert_close(cs, cref, 1e-5, 1e-4 * np.sqrt(K), f"colsum {i}")


# ---------------------------------------------------------------------------------------------------------------
# grid-stride launches (more work items than resident workgroups: the grid is a multiple of the resident count)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,B,H,W,h,w", [(2, 9000, 50, 50, 20, 20), (3, 5000, 28, 36, 9, 12), (1, 7001, 17, 13, 5, 7)])
def test_grid_stride_launches_equal_small_launches_bit_for_bit(hip, T, B, H, W, h, w):
    """Every ST / canvas launch whose items outnumber the resident workgroups (read forward in both forms, read backward, canvas
    forward, stored-canvas and recompute backward) against the SAME kernels launched on chunks of 600 images (one item per
    workgroup, the same 256-thread workgroup shape): a unit's arithmetic does not depend on which workgroup of which grid works on it,
    so the bits must agree."""
    gen = torch.Generator(device="cuda").manual_seed(T * 1000 + B)
    dev = torch.device("cuda")
    img = torch.rand(B, H, W, device=dev, generator=gen)
    where = torch.empty(T, B, 4, device=dev)
    where[..., 0] = 0.3 + 0.9 * torch.rand(T, B, device=dev, generator=gen)
    where[..., 2] = 0.3 + 0.9 * torch.rand(T, B, device=dev, generator=gen)
    where[..., 1] = 1.2 * torch.rand(T, B, device=dev, generator=gen) - 0.6
    where[..., 3] = 1.2 * torch.rand(T, B, device=dev, generator=gen) - 0.6
    pres = (torch.rand(T, B, device=dev, generator=gen) < 0.7).float()
    glm = torch.randn(T, B, h, w, device=dev, generator=gen)
    dgl = torch.randn(T * B, h, w, device=dev, generator=gen)
    CH = 600

    def chunks():          # every chunk holds 600 .. 1199 images: the 256-thread, one-band workgroup shape of the large launch
        b0 = 0
        while B - b0 >= 2 * CH:
            yield b0, b0 + CH
            b0 += CH
        yield b0, B

    # read forward: T glimpses per staged image (the vectorised / lean forms), and one image per glimpse
    out = hip.st_read_fwd(img, where.reshape(T * B, 4), (h, w), n_img=B).view(T, B, h, w)
    for b0, b1 in chunks():
        ref = hip.st_read_fwd(img[b0:b1].contiguous(), where[:, b0:b1].reshape(-1, 4).contiguous(), (h, w), n_img=b1 - b0)
        assert torch.equal(out[:, b0:b1], ref.view(T, b1 - b0, h, w)), f"read forward, images {b0}..{b1}"
    out1 = hip.st_read_fwd(img, where[0].contiguous(), (h, w))
    for b0, b1 in chunks():
        assert torch.equal(out1[b0:b1], hip.st_read_fwd(img[b0:b1].contiguous(), where[0, b0:b1].contiguous(), (h, w)))
    # read backward (dwhere): one workgroup per image beyond 2048 glimpses
    dwh, _ = hip.st_read_bwd(img, where.reshape(T * B, 4), dgl)
    for b0, b1 in chunks():
        ref, _ = hip.st_read_bwd(img[b0:b1].contiguous(), where[:, b0:b1].reshape(-1, 4).contiguous(),
                                 dgl.view(T, B, h, w)[:, b0:b1].reshape(-1, h, w).contiguous())
        assert torch.equal(dwh.view(T, B, 4)[:, b0:b1], ref.view(T, b1 - b0, 4)), f"read backward, images {b0}..{b1}"
    # canvas forward, stored-canvas backward, recompute backward
    steps, final, rec = hip.canvas_unroll_fwd(glm, where, pres, (H, W), obs=img, mult=1.0, std=0.3)
    dg_s, dw_s = hip.canvas_unroll_bwd(glm, where, pres, img, final, 1.0, 0.3, 1.0 / B)
    dg_r, dw_r = hip.canvas_unroll_bwd(glm, where, pres, img, None, 1.0, 0.3, 1.0 / B)
    for b0, b1 in chunks():
        gl_c, wh_c, pr_c = glm[:, b0:b1].contiguous(), where[:, b0:b1].contiguous(), pres[:, b0:b1].contiguous()
        st_c, fi_c, re_c = hip.canvas_unroll_fwd(gl_c, wh_c, pr_c, (H, W), obs=img[b0:b1].contiguous(), mult=1.0, std=0.3)
        assert torch.equal(steps[:, b0:b1], st_c) and torch.equal(final[b0:b1], fi_c), f"canvas forward, images {b0}..{b1}"
        assert torch.equal(rec[b0:b1], re_c), f"reconstruction term, images {b0}..{b1}"          # (same workgroup shape: same sum order)
        for name, (dg, dw), fc in (("stored", (dg_s, dw_s), fi_c), ("recompute", (dg_r, dw_r), None)):
            dg_c, dw_c = hip.canvas_unroll_bwd(gl_c, wh_c, pr_c, img[b0:b1].contiguous(), fc, 1.0, 0.3, 1.0 / B)
            assert torch.equal(dg[:, b0:b1], dg_c) and torch.equal(dw[:, b0:b1], dw_c), f"canvas backward ({name}), images {b0}..{b1}"


@pytest.mark.parametrize("B,T,H,W,h,w", [(704, 3, 50, 50, 20, 20), (416, 5, 100, 100, 28, 28), (688, 3, 12, 10, 3, 4)])
def test_attend_fwd_image_major_lean_read_is_bit_identical(hip, monkeypatch, B, T, H, W, h, w):
    """air_attend_fwd beyond 2048 glimpses runs one workgroup per image; round 6 resamples the T glimpses there as the lean read kernel does
    (bordered image in LDS, axis tables, four outputs per thread).  Every output equals the per-pixel form's bit for bit, and the glimpses
    equal the CPU oracle's read of the same images with the `where` rows the launch sampled (modules.py:94-109, cell.py:129-135)."""
    import ctypes
    from attend_infer_repeat_amd import _lib
    lib = hip.lib()
    rng = np.random.default_rng(B + T)
    M, Kt, Ks = T * B, 64, 32
    f = lambda *shape, s=1.0: g((s * rng.standard_normal(shape)).astype(np.float32))
    img = g((rng.random((B, H, W)) * (rng.random((B, H, W)) < 0.3)).astype(np.float32))
    tr_h, tr_w, tr_b = f(M, Kt), f(Kt, 8, s=0.2), f(8, s=0.5)
    st_h, st_w, st_b = f(M, Ks), f(Ks, 1, s=0.2), f(1)
    eps, u = f(M, 4), g(rng.random(M).astype(np.float32))
    prior = g(np.array([0.5 ** (n + 1) for n in range(T + 1)]), torch.float64)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(lean):
        if lean:
            monkeypatch.delenv("AIR_ATTEND_LEAN", raising=False)
        else:
            monkeypatch.setenv("AIR_ATTEND_LEAN", "0")
        z = lambda *shape: torch.full(shape, float("nan"), device="cuda")
        out = dict(pre=z(M, 8), logit=z(M), loc=z(M, 4), scale=z(M, 4), where=z(M, 4), kl=z(M), prob=z(M), pres=z(M), q=z(B, T + 1),
                   klps=z(B), logp=z(B), stepw=z(M), glimpse=z(M, h * w))
        _lib.check(lib.air_attend_fwd(p(tr_h), p(tr_w), p(tr_b), Kt, p(st_h), p(st_w), p(st_b), Ks, p(out["pre"]), p(out["logit"]), p(eps),
                                      0.5, 0.0, 1.0, 0.0, 1.0, p(out["loc"]), p(out["scale"]), p(out["where"]), p(out["kl"]), p(u), 0.0, 0.0,
                                      p(prior), p(out["prob"]), p(out["pres"]), p(out["q"]), p(out["klps"]), p(out["logp"]), p(out["stepw"]),
                                      p(img), p(out["glimpse"]), T, B, H, W, h, w, 0, 1e-3, hip._stream()), "air_attend_fwd")
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}

    a, b = run(True), run(False)
    for k in a:
        assert not np.isnan(a[k]).any(), k
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    ref = C.st_read_fwd(np.tile(img.cpu().numpy(), (T, 1, 1)), a["where"], (h, w))
    np.testing.assert_array_equal(a["glimpse"].reshape(M, h, w), ref)
