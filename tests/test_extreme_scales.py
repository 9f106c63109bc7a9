"""The regime where long training runs die: the transform head's raw scale wanders to
-44 ... -100, sigma = softplus(raw) reaches 1e-20 ... 0, sigma^2 turns denormal and then underflows.  In the reference's fp32
arithmetic (model.py:188-214 -- the Normal KL is written over ratio = sigma^2 / prior^2 and differentiated as written,
cell.py:130-133 / modules.py:62-63 -- softplus without a floor) the KL row is then +inf and the scale gradient -inf or NaN.
SURVEY section 7: "match, don't clamp".  These tests hold the HIP kernels to the SAME inf / NaN placement as the fp32 oracle,
element for element, and to its values wherever both are finite:

  * kernel level (air_gauss_sample_fwd/bwd, air_normal_kl_fwd/bwd): exact inputs on both sides;
  * engine level (air_attend_fwd / air_attend_bwd / canvas kernels / the whole backward): the transform head's (and the `what`
    head's) scale biases are set to the extreme values, |where| reaches +-250, and every output and gradient tensor must be
    finite, +inf, -inf or NaN exactly where the fp32 oracle's is.
"""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import air_oracle as O

pytestmark = pytest.mark.gpu

# raw scales: ordinary | sigma^2 denormal | sigma^2 = 0 (KL = +inf, gradient -inf) | exp(-raw) = inf (gradient inf/inf = NaN) |
# sigma denormal | sigma = 0
RAWS = [86.0, 21.0, 20.0, 0.0, -13.0, -20.0, -30.0, -43.0, -44.0, -46.0, -51.0, -53.0, -60.0, -88.0, -89.0, -103.0, -104.0,
        -110.0, -200.0]


def classes(t):
    """0 finite, 1 +inf, 2 -inf, 3 NaN"""
    t = t.detach().cpu()
    c = torch.zeros(t.shape, dtype=torch.int8)
    c[torch.isposinf(t)] = 1
    c[torch.isneginf(t)] = 2
    c[torch.isnan(t)] = 3
    return c


def same_placement(name, got, ref):
    cg, cr = classes(got), classes(ref.reshape(got.shape))
    bad = (cg != cr).nonzero()
    assert bad.numel() == 0, "%s: %d elements differ in class (0 finite, 1 +inf, 2 -inf, 3 NaN); first at %s: kernel %d, oracle %d" % (
        name, bad.shape[0], bad[0].tolist(), int(cg[tuple(bad[0])]), int(cr[tuple(bad[0])]))
    return cr == 0


def close_where_finite(name, got, ref, tol, floor=0.0, abs_tol=0.0):
    """same classes everywhere; where finite: |got - ref| <= tol * (|ref| + floor) + abs_tol"""
    fin = same_placement(name, got, ref)
    g, r = got.detach().cpu().double().reshape(fin.shape)[fin], ref.detach().cpu().double().reshape(fin.shape)[fin]
    if g.numel():
        err = ((g - r).abs() / (tol * (r.abs() + floor) + abs_tol)).max().item()
        worst = int(((g - r).abs() / (tol * (r.abs() + floor) + abs_tol)).argmax())
        assert err < 1.0, (name, err, float(g[worst]), float(r[worst]))


@pytest.mark.parametrize("D,loc_mode,offset", [(4, 1, 0.5), (50, 0, 0.5)])
def test_gauss_sample_kl_extreme_scales_same_placement_as_fp32_oracle(gpu_device, D, loc_mode, offset):
    """air_gauss_sample_fwd / _bwd (cell.py:130-133,154-156 + the KL rows of model.py:177-214) on rows whose raw scale walks
    through RAWS: sample, scale, KL row and d(pre) against the fp32 oracle on the SAME inputs."""
    from attend_infer_repeat_amd import hip as H
    rng = np.random.default_rng(7)
    M = len(RAWS) * 3
    pre = torch.from_numpy(rng.standard_normal((M, 2 * D)).astype(np.float32))
    for m in range(M):
        pre[m, D:] = RAWS[m % len(RAWS)] - offset          # whole row at one raw value ...
    for m in range(len(RAWS), 2 * len(RAWS)):              # ... or one column only (the rest of the row ordinary)
        keep = pre[m, D].clone(); pre[m, D:] = torch.from_numpy(rng.standard_normal(D).astype(np.float32)); pre[m, D] = keep
    eps = torch.from_numpy(rng.standard_normal((M, D)).astype(np.float32))
    dsample = torch.from_numpy(rng.standard_normal((M, D)).astype(np.float32))
    dkl = torch.from_numpy((rng.uniform(1e-6, 1e-2, M)).astype(np.float32))
    dkl[2 * len(RAWS):] = 0.0                              # third block: zero KL weight (0 * inf = NaN where the reference's is)
    prior4 = (0.0, 1.0, 0.0, 1.0)

    p = pre.clone().requires_grad_(True)
    mu, raw = p[:, :D], p[:, D:]
    if loc_mode == 1:
        ocfg = O.AIRConfig(transform_var_bias=offset)
        mu, raw = O.transform_params(p, ocfg)
    else:
        raw = raw + offset
    sc = O.softplus(raw)
    smp = mu + sc * eps
    one = torch.ones(())
    kl = O.normal_kl(mu, sc, 0.0 * one, one).sum(-1)
    ((smp * dsample).sum() + (kl * dkl).sum()).backward()

    loc_d, scale_d, sample_d, kl_d = H.gauss_sample_fwd(pre.cuda(), eps.cuda(), offset, loc_mode, prior4)
    close_where_finite("loc", loc_d, mu, 2e-6, 1e-30)
    # softplus of a raw below -87 is a denormal on both sides: a handful of bits, compared to an ulp of the denormal grid
    close_where_finite("scale", scale_d, sc, 4e-6, 0.0, abs_tol=3e-45)      # (denormal results: two steps of the denormal grid)
    close_where_finite("sample", sample_d, smp, 1e-5, 0.0, abs_tol=1e-6)       # (loc + scale * eps cancels: errors scale with the operands)
    same_placement("kl_row", kl_d, kl)
    assert torch.isposinf(kl_d.cpu()[[m for m in range(M) if RAWS[m % len(RAWS)] <= -53.0 and m < len(RAWS)]]).all()
    dpre = H.gauss_sample_bwd(pre.cuda(), eps.cuda(), offset, loc_mode, prior4, loc_d, scale_d, dsample.cuda(), dkl.cuda())
    fin = same_placement("dpre", dpre, p.grad)
    # values: the scale handed to the backward is the kernel's own (an ulp from the oracle's); where sigma^2 is denormal an ulp of
    # sigma moves 1 / ratio by up to a few per cent, so those rows are held to placement + sign + 10 %, everything else to 2e-4
    g, r = dpre.cpu().double(), p.grad.double()
    den = torch.zeros(M, dtype=torch.bool)
    for m in range(M):
        den[m] = -51.5 < RAWS[m % len(RAWS)] < -43.5
    for rows, tol in ((~den, 2e-4), (den, 0.1)):
        sel = fin & rows[:, None]
        if sel.any():
            err = ((g[sel] - r[sel]).abs() / (r[sel].abs() + 1e-6 * r[fin].abs().max())).max().item()
            assert err < tol, (tol, err)


def test_normal_kl_extreme_scales_same_placement_as_fp32_oracle(gpu_device):
    """air_normal_kl_fwd / _bwd with the scale given directly (bit-identical inputs on both sides): every class boundary of
    ratio = s^2 in fp32 -- last normal, denormal, smallest denormal, underflow -- and of 0.5 g / ratio overflowing."""
    from attend_infer_repeat_amd import hip as H
    scales = [86.0, 1.0, 0.3, 1e-9, 1.1e-19, 1.0e-19, 7.8e-20, 1e-20, 1e-21, 2e-22, 1e-22, 5e-23, 3.8e-23, 3.0e-23, 1e-23, 1e-30,
              1.2e-38, 1e-38, 1e-42, 1.4e-45, 0.0]
    M, D = len(scales), 4
    rng = np.random.default_rng(11)
    loc = torch.from_numpy(rng.standard_normal((M, D)).astype(np.float32))
    scale = torch.tensor(scales, dtype=torch.float32)[:, None].repeat(1, D).contiguous()
    dkl = torch.from_numpy(rng.uniform(1e-6, 1e-2, M).astype(np.float32))
    prior4 = (0.0, 1.0, 0.5, 2.0)
    lo, sc = loc.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    pm = torch.tensor([0.0, 0.5, 0.0, 0.5]); ps = torch.tensor([1.0, 2.0, 1.0, 2.0])
    kl = O.normal_kl(lo, sc, pm, ps).sum(-1)
    (kl * dkl).sum().backward()
    kl_d = H.normal_kl_fwd(loc.cuda(), scale.cuda(), prior4)
    close_where_finite("kl", kl_d, kl, 1e-5, 1e-6)
    dloc, dscale = H.normal_kl_bwd(loc.cuda(), scale.cuda(), prior4, dkl.cuda())
    close_where_finite("dloc", dloc, lo.grad, 1e-5, 1e-9)
    close_where_finite("dscale", dscale, sc.grad, 1e-5, 1e-9)
    c = classes(dscale)
    assert (c[scales.index(1e-30)] == 2).all() and (c[scales.index(0.0)] == 3).all() and (c[scales.index(1.0e-19)] == 0).all()


# (name, raw scale of [sx, tx, sy, ty], raw scale of the first `what` columns or None)
ENGINE_CASES = [
    ("sigma2_normal", (-20.0, -30.0, -40.0, -43.0), None),           # everything finite, sigma^2 down to the last normal decade
    ("sigma2_denormal_and_wide", (-44.0, -46.0, 20.0, 86.0), None),   # sigma^2 denormal; scale 86 => |where| up to +-250
    ("sigma2_underflow", (-60.0, 0.0, -88.0, 0.0), None),            # KL = +inf, scale gradient -inf
    ("sigma_zero", (-89.0, -104.0, -110.0, -200.0), None),           # -inf / inf and -inf * 0 = NaN
    ("what_underflow", (0.0, 0.0, 0.0, 0.0), (-60.0, -104.0, -44.0)),
]


@pytest.mark.parametrize("case", ENGINE_CASES, ids=[c[0] for c in ENGINE_CASES])
@pytest.mark.parametrize("B", [8, 64])
def test_engine_extreme_scales_same_placement_as_fp32_oracle(gpu_device, case, B):
    """The whole train-step arithmetic (air_attend_fwd / air_attend_bwd with the where-sampling backward on 4 lanes, the canvas
    kernels, air_gauss_sample_bwd_nvil, every GEMM behind them) with the scale heads forced to the extreme regime: outputs, losses
    and ALL gradient tensors are finite / +inf / -inf / NaN exactly where the fp32 oracle's are, and equal where both are finite."""
    from test_engine import make_pair
    name, where_raw, what_raw = case
    ocfg = O.AIRConfig()
    eng, params, obs, noise = make_pair(ocfg, B, seed=5)
    params = {k: v.clone() for k, v in params.items()}
    last = "transform/%d" % len(ocfg.transform_estimator_hidden)
    params[last + "/w"][:, 4:] = 0.0                                    # the raw scales come from the bias alone: every row the same
    params[last + "/b"][4:] = torch.tensor(where_raw) - ocfg.transform_var_bias
    if what_raw is not None:
        A, n = ocfg.n_appearance, len(what_raw)
        params["what/w"][:, A:A + n] = 0.0
        params["what/b"][A:A + n] = torch.tensor(what_raw) - ocfg.what_scale_offset
    eng.load_parameters(params)
    eng.forward(sample_noise=False); eng.backward()
    out, grads = eng.outputs(), eng.named_grads()
    res, ref = O.forward_backward(params, ocfg, obs, noise, global_step=20000)          # fp32: placement is a property of fp32

    assert torch.isfinite(out["where"]).all() and torch.isfinite(res["where"]).all()
    if name == "sigma2_denormal_and_wide":
        assert out["where"].abs().max().item() > 140.0
    for k in ("where", "where_loc", "what_loc", "presence_prob", "final_canvas", "rec_loss_per_sample",
              "kl_num_steps_per_sample"):
        close_where_finite(k, out[k], res[k], 2e-3, res[k].abs().max().item() + 1e-30)      # (of the tensor's largest element)
    for k in ("where_scale", "what_scale"):
        close_where_finite(k, out[k], res[k], 1e-4, 0.0, abs_tol=3e-45)
    for k in ("kl_where_per_sample", "kl_what_per_sample", "loss", "opt_loss", "kl_where", "kl_what"):
        fin = same_placement(k, out[k], res[k])
        if name in ("sigma2_underflow", "sigma_zero") and k in ("kl_where_per_sample", "kl_where", "loss", "opt_loss"):
            assert torch.isposinf(out[k]).all(), k
        if name == "what_underflow" and k in ("kl_what_per_sample", "kl_what", "loss", "opt_loss"):
            assert torch.isposinf(out[k]).all(), k
        if fin.all():
            close_where_finite(k, out[k], res[k], 1e-3, 1e-3 * res[k].abs().max().item() + 1e-30)
    n_bad = 0
    for k, r in ref.items():
        fin = same_placement("grad " + k, grads[k], r)
        n_bad += int((~fin).sum())
        if fin.any():
            g, rr = grads[k].detach().cpu().double()[fin], r.double()[fin]
            # denormal sigma^2: an ulp of sigma moves the gradient's 1 / ratio term by per cents (see the kernel-level test)
            tol = 0.1 if name in ("sigma2_denormal_and_wide", "what_underflow") else 2e-3
            err = ((g - rr).abs().max() / (rr.abs().max() + 1e-30)).item()
            assert err < tol, (k, err)
    if name == "sigma2_normal" or (name == "sigma2_denormal_and_wide" and B >= 64):
        assert n_bad == 0          # (at batch 8 the KL weight 1/B is large enough for .5 dk / ratio to overflow on a denormal ratio)
    elif name != "sigma2_denormal_and_wide":
        assert n_bad > 0
        # the networks downstream of the latents never see the scale gradient: they stay finite while everything upstream of the
        # transform (or `what`) head is poisoned -- in the oracle and in the engine alike.  (The steps predictor is poisoned too:
        # the step weights q(n > t) multiply the KL rows, so an infinite row is an infinite d loss / d presence_prob.)
        for k in ref:
            if k.startswith("baseline/") or k.startswith("glimpse_decoder/"):
                assert torch.isfinite(grads[k]).all(), k
        assert not torch.isfinite(grads[(last if what_raw is None else "what") + "/b"]).all()
    # one update: RMSProp carries any non-finite gradient into the parameter (ms - mg^2 = inf - inf), as DESIGN section 6 describes
    eng.optimizer_step(); eng.synchronize()
    got_finite = bool(torch.isfinite(eng.flat_params).all())
    slots = O.rmsprop_init(params)
    O.rmsprop_centered_step(params, ref, slots, ocfg)
    ref_finite = all(bool(torch.isfinite(v).all()) for v in params.values())
    assert got_finite == ref_finite == (n_bad == 0)
    for k, v in params.items():
        same_placement("updated " + k, eng.params[k], v)


GUARD = 1e-6
GUARD32 = float(np.float32(GUARD))          # the floor as the fp32 kernels (and the fp32 oracle) hold it


@pytest.mark.parametrize("case", ENGINE_CASES, ids=[c[0] for c in ENGINE_CASES])
@pytest.mark.parametrize("B", [8, 64])
def test_engine_extreme_scales_with_the_guard_stay_finite_and_match_the_guarded_oracle(gpu_device, case, B):
    """the documented stability switch (EngineConfig.guard_eps / AIRonMNIST(guard_degenerate=...), default off).
    The same extreme scale heads as above with guard_eps = 1e-6: every output, loss and gradient tensor is FINITE, the scales sit
    on the floor, and everything equals the oracle's restatement of the guard (O.AIRConfig.guard_eps) -- including one update."""
    from test_engine import make_pair
    name, where_raw, what_raw = case
    ocfg = O.AIRConfig(guard_eps=GUARD)
    eng, params, obs, noise = make_pair(ocfg, B, seed=5)
    assert eng.cfg.guard_eps == GUARD
    params = {k: v.clone() for k, v in params.items()}
    last = "transform/%d" % len(ocfg.transform_estimator_hidden)
    params[last + "/w"][:, 4:] = 0.0
    params[last + "/b"][4:] = torch.tensor(where_raw) - ocfg.transform_var_bias
    if what_raw is not None:
        A, n = ocfg.n_appearance, len(what_raw)
        params["what/w"][:, A:A + n] = 0.0
        params["what/b"][A:A + n] = torch.tensor(what_raw) - ocfg.what_scale_offset
    eng.load_parameters(params)
    eng.forward(sample_noise=False); eng.backward()
    out, grads = eng.outputs(), eng.named_grads()
    res, ref = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    for k, v in out.items():
        if torch.is_tensor(v):
            assert torch.isfinite(v).all(), k
    assert out["where_scale"].min().item() >= GUARD32 and out["what_scale"].min().item() >= GUARD32
    if name in ("sigma2_underflow", "sigma_zero"):
        assert (out["where_scale"] == GUARD32).any()                     # the floor is what these heads sit on
    for k in ("where", "where_loc", "where_scale", "what_loc", "what_scale", "presence_prob", "final_canvas", "rec_loss_per_sample",
              "kl_num_steps_per_sample", "kl_where_per_sample", "kl_what_per_sample"):
        close_where_finite(k, out[k], res[k], 2e-3, res[k].abs().max().item() + 1e-30)
    for k in ("loss", "opt_loss", "kl_where", "kl_what"):
        close_where_finite(k, out[k], res[k], 1e-3, 1e-3 * res[k].abs().max().item() + 1e-30)
    for k, r in ref.items():
        assert torch.isfinite(grads[k]).all() and torch.isfinite(r).all(), k
        err = ((grads[k].cpu().double() - r.double()).abs().max() / (r.double().abs().max() + 1e-30)).item()
        assert err < 2e-3, (k, err)
    # the raw-scale gradient of a floored head is exactly zero (no gradient through the floor), on both sides
    if name in ("sigma2_underflow", "sigma_zero"):
        floored = (res["where_scale"].reshape(-1, 4) == GUARD32).all(0)
        assert floored.any()
        assert (grads[last + "/b"].cpu()[4:][floored] == 0).all() and (ref[last + "/b"][4:][floored] == 0).all()
    eng.optimizer_step(); eng.synchronize()
    assert torch.isfinite(eng.flat_params).all()


def test_where_scale_sample_is_kept_off_zero_by_the_guard(gpu_device):
    """cell.py:130-133 can sample an exact zero scale (where = loc + scale * eps cancels to the last bit; seed 9 of round 4 died of
    it at update 1320): with the guard the sampled scale components keep |s| >= guard_eps (sign kept, +guard for +0), the shift
    components are untouched, and without it the sample is the reference's."""
    from attend_infer_repeat_amd import hip as H
    pre = torch.zeros(6, 8)
    pre[:, 4:] = 0.5413                                                   # softplus(raw) ~ 1
    eps = torch.zeros(6, 4)
    sig = torch.sigmoid(torch.zeros(())).item()                           # loc of a scale component = 0.5
    sc = torch.nn.functional.softplus(torch.tensor(0.5413)).item()
    eps[0, 0] = -sig / sc                                                 # cancels (to rounding)
    eps[1, 2] = -sig / sc
    eps[2, 1] = 0.0                                                       # a shift component at tanh(0) = 0 stays 0
    loc, scale, samp, _ = H.gauss_sample_fwd(pre.cuda(), eps.cuda(), 0.0, 1, (0., 1., 0., 1.), want_kl=False, guard_eps=1e-3)
    ref_loc, ref_scale, ref_samp, _ = H.gauss_sample_fwd(pre.cuda(), eps.cuda(), 0.0, 1, (0., 1., 0., 1.), want_kl=False)
    samp, ref_samp = samp.cpu(), ref_samp.cpu()
    assert ref_samp[0, 0].abs().item() < 1e-6 and ref_samp[1, 2].abs().item() < 1e-6
    assert samp[0, 0].abs().item() == pytest.approx(1e-3) and samp[1, 2].abs().item() == pytest.approx(1e-3)
    assert samp[2, 1].item() == 0.0 and samp[:, 1::2].equal(ref_samp[:, 1::2])
    keep = ref_samp.abs() >= 1e-3
    assert samp[keep].equal(ref_samp[keep])


# sampled scales of the inverse warp that break 1/s, 1/s^2 or s^2 in fp32 (sx, sy are unbounded samples; the
# reference divides by them, modules.py:101-102).  An exact 0.0 is not exotic: where = loc + scale * eps cancels to the last bit with
# probability ~1e-8 per draw while the scale is O(1) -- profiles/r04_blowup_seed9_legacy.json is such an update (update 1320).
DEGENERATE = [0.0, -0.0, 1e-40, -1e-40, 1e-30, 1e-20, -1e-20, 1e19, -3e38, 3e38]


@pytest.mark.parametrize("form", ["given_dcanvas", "stored_canvas", "recompute"])
def test_canvas_write_at_degenerate_scales_same_placement_as_fp32_oracle(gpu_device, form):
    """air_st_write_fwd / _bwd and air_canvas_unroll_fwd / _bwd (cell.py:159-165) with sx or sy at DEGENERATE: the forward is the
    oracle's bit for bit (a glimpse that lands nowhere writes zeros), and dwhere / dglimpse are NaN / inf / finite exactly where the
    fp32 oracle's are -- the reference's inverse warp has no guard, so neither has this one."""
    from attend_infer_repeat_amd import hip as H
    rng = np.random.default_rng(3)
    rows = []
    for v in DEGENERATE:
        rows += [[v, 0.3, 0.7, -0.2], [0.6, -0.1, v, 0.25], [v, 0.0, v, 0.0]]
    rows += [[0.5, 0.1, 0.5, 0.1], [1.0, 0.0, 1.0, 0.0]]
    n, (Hc, Wc), (h, w) = len(rows), (50, 50), (20, 20)
    where = torch.tensor(rows, dtype=torch.float32)
    glm = torch.from_numpy(rng.standard_normal((n, h, w)).astype(np.float32))
    pres = torch.from_numpy((rng.random(n) < 0.7).astype(np.float32)); pres[-2:] = 1.0
    if form == "given_dcanvas":
        dcan = torch.from_numpy(rng.standard_normal((n, Hc, Wc)).astype(np.float32))
        wr, gl = where.clone().requires_grad_(True), glm.clone().requires_grad_(True)
        out = O.st_write(gl, wr, (Hc, Wc)) * pres[:, None, None]
        (out * dcan).sum().backward()
        fwd = H.st_write_fwd(glm.cuda(), where.cuda(), (Hc, Wc), presence=pres.cuda())
        assert torch.equal(fwd.cpu(), out.detach())                         # bit for bit, zeros where the warp is degenerate
        dg, dwh, _ = H.st_write_bwd(glm.cuda(), where.cuda(), dcan.cuda(), presence=pres.cuda())
    else:
        # T = 1 canvases (every row its own image): the reconstruction term drives dcanvas
        obs = torch.from_numpy(rng.random((n, Hc, Wc)).astype(np.float32))
        mult, std, scale = 0.5, 0.3, 1.0 / n
        wr, gl = where.clone().requires_grad_(True), glm.clone().requires_grad_(True)
        canvas = O.st_write(gl, wr, (Hc, Wc)) * pres[:, None, None]
        nll = 0.5 * ((obs - mult * canvas) / std) ** 2
        (nll.sum() * scale).backward()
        steps, final, _ = H.canvas_unroll_fwd(glm.cuda()[None], where.cuda()[None], pres.cuda()[None], (Hc, Wc), obs=obs.cuda(),
                                              mult=mult, std=std)
        assert torch.equal(final.cpu(), canvas.detach())
        dg, dwh = H.canvas_unroll_bwd(glm.cuda()[None], where.cuda()[None], pres.cuda()[None], obs.cuda(),
                                      final if form == "stored_canvas" else None, mult, std, scale)
        dg, dwh = dg[0], dwh[0]
    same_placement("dglimpse", dg, gl.grad)
    fin = same_placement("dwhere", dwh, wr.grad)
    assert not fin.all() and fin[-2:].all()                                # the degenerate rows do produce non-finite gradients
    g, r = dwh.cpu().double()[fin], wr.grad.double()[fin]
    assert ((g - r).abs() <= 2e-4 * (r.abs() + 1e-3 * wr.grad[-2:].abs().max().item())).all()
    gfin = torch.isfinite(gl.grad)
    assert ((dg.cpu().double()[gfin] - gl.grad.double()[gfin]).abs() <= 2e-4 * gl.grad[gfin].abs().max().item() + 1e-12).all()


def test_glimpse_read_at_degenerate_scales_matches_fp32_oracle(gpu_device):
    """the read direction has no division (modules.py:104-109): sx = 0 reads one column, huge scales read nothing; everything finite"""
    from attend_infer_repeat_amd import hip as H
    rng = np.random.default_rng(5)
    rows = [[v, 0.3, 0.7, -0.2] for v in DEGENERATE] + [[0.6, -0.1, v, 0.25] for v in DEGENERATE]
    n = len(rows)
    where = torch.tensor(rows, dtype=torch.float32)
    img = torch.from_numpy(rng.random((n, 50, 50)).astype(np.float32))
    dgl = torch.from_numpy(rng.standard_normal((n, 20, 20)).astype(np.float32))
    wr = where.clone().requires_grad_(True)
    out = O.st_read(img, wr, (20, 20))
    (out * dgl).sum().backward()
    got = H.st_read_fwd(img.cuda(), where.cuda(), (20, 20))
    assert torch.equal(got.cpu(), out.detach())
    dwh = H.st_read_bwd(img.cuda(), where.cuda(), dgl.cuda())
    dwh = dwh[0] if isinstance(dwh, (tuple, list)) else dwh
    fin = same_placement("dwhere", dwh, wr.grad)
    assert fin.all()
    assert ((dwh.cpu().double() - wr.grad.double()).abs() <= 2e-4 * (wr.grad.abs().double() + 1e-3 * wr.grad.abs().max().item())).all()
