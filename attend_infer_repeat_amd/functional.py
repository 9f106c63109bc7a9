"""torch.autograd.Function wrappers over the HIP kernels (hip.py -> include/air_hip.h).

Each Function is one forward kernel + its hand-written backward kernel(s); autograd only orders them.  These are what
the drop-in modules (neural.py / modules.py / cell.py / model.py) are made of when a model is stepped cell by cell or
uses a non-standard architecture; the standard architecture trains through engine.AIREngine instead.
"""
import torch

from . import hip as H

_c = lambda t: t if t is None or t.is_contiguous() else t.contiguous()


class _Linear(torch.autograd.Function):
    """y = act(x.w + b)   (Affine, neural.py:56-60)"""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w, b = _c(x), _c(w), _c(b)
        y = H.linear_fwd(x, w, b, act)
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.has_b = act, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dx, dw, db = H.linear_bwd(x, w, y, _c(dy), ctx.act, want_dx=ctx.needs_input_grad[0], want_db=ctx.has_b)
        return dx, dw, db, None


def linear(x, w, b=None, act=H.ACT_NONE):
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), w, b, act)
    return y.reshape(*lead, w.shape[1])


class _StRead(torch.autograd.Function):
    """glimpse = resample(img, grid(where))   (SpatialTransformer, modules.py:104-109)"""

    @staticmethod
    def forward(ctx, img, where, crop_size):
        img, where = _c(img), _c(where)
        out = H.st_read_fwd(img, where, crop_size)
        ctx.save_for_backward(img, where)
        return out

    @staticmethod
    def backward(ctx, dout):
        img, where = ctx.saved_tensors
        dwhere, dimg = H.st_read_bwd(img, where, _c(dout), want_dimg=ctx.needs_input_grad[0])
        return dimg, dwhere, None


def st_read(img, where, crop_size):
    return _StRead.apply(img, where, tuple(crop_size))


class _StWrite(torch.autograd.Function):
    """inverse warp of a glimpse onto the canvas (SpatialTransformer(inverse=True), modules.py:101-102,109)"""

    @staticmethod
    def forward(ctx, glimpse, where, img_size):
        glimpse, where = _c(glimpse), _c(where)
        out = H.st_write_fwd(glimpse, where, img_size)
        ctx.save_for_backward(glimpse, where)
        return out

    @staticmethod
    def backward(ctx, dout):
        glimpse, where = ctx.saved_tensors
        dg, dwhere, _ = H.st_write_bwd(glimpse, where, _c(dout))
        return dg, dwhere, None


def st_write(glimpse, where, img_size):
    return _StWrite.apply(glimpse, where, tuple(img_size))


class _StWriteAcc(torch.autograd.Function):
    """canvas_out = canvas_in + presence * inverse_warp(glimpse, where)   (cell.py:159-165, one fused kernel)"""

    @staticmethod
    def forward(ctx, glimpse, where, presence, canvas_in, img_size):
        glimpse, where, presence, canvas_in = _c(glimpse), _c(where), _c(presence), _c(canvas_in)
        out = H.st_write_fwd(glimpse, where, img_size, presence=presence.reshape(-1), canvas_in=canvas_in)
        ctx.save_for_backward(glimpse, where, presence)
        return out

    @staticmethod
    def backward(ctx, dout):
        glimpse, where, presence = ctx.saved_tensors
        dout = _c(dout)
        dg, dwhere, dpres = H.st_write_bwd(glimpse, where, dout, presence=presence.reshape(-1),
                                           want_dpresence=ctx.needs_input_grad[2])
        return dg, dwhere, (dpres.reshape(presence.shape) if dpres is not None else None), dout, None


def st_write_acc(glimpse, where, presence, canvas_in, img_size):
    return _StWriteAcc.apply(glimpse, where, presence, canvas_in, tuple(img_size))


class _Nvil(torch.autograd.Function):
    """reinforce_loss, baseline_loss (+ imp-weight mean/var) with the reference's [B]-[B,1] broadcast (model.py:218-259).
    Gradients: reinforce_loss -> logp only (importance weight is stop_gradient'ed); baseline_loss -> baseline only."""

    @staticmethod
    def forward(ctx, imp, baseline, logp):
        out, dlogp, dbase = H.nvil(_c(imp.reshape(-1)), _c(baseline.reshape(-1)), _c(logp.reshape(-1)))
        ctx.save_for_backward(dlogp, dbase)
        ctx.bshape = baseline.shape
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, d_rl, d_bl, _dm, _dv):
        dlogp, dbase = ctx.saved_tensors
        return None, (dbase * d_bl).reshape(ctx.bshape), dlogp * d_rl


def nvil(imp, baseline, logp):
    return _Nvil.apply(imp, baseline, logp)


class _LstmPointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, c_prev, forget_bias):
        gates, c_prev = _c(gates), _c(c_prev)
        h, c, act = H.lstm_pointwise_fwd(gates, c_prev, forget_bias)
        ctx.save_for_backward(act, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        act, c_prev, c = ctx.saved_tensors
        dgates, dc_prev = H.lstm_pointwise_bwd(act, c_prev, c, _c(dh), _c(dc))
        return dgates, dc_prev, None


def lstm_cell(x, h, c, w_gates, b_gates, forget_bias=1.0):
    """Sonnet v1 LSTM step (mnist_model.py:35, cell.py:126-127): gates = [x,h].W + b; i,j,f,o."""
    I = x.shape[-1]
    gx = linear(x, w_gates[:I], b_gates)
    gates = gx + linear(h, w_gates[I:], None)
    return _LstmPointwise.apply(gates, c, float(forget_bias))


class _GaussSample(torch.autograd.Function):
    """loc, scale=softplus(raw+offset), sample=loc+scale*eps from pre=[loc_pre | raw]   (cell.py:130-133,154-156)"""

    @staticmethod
    def forward(ctx, pre, eps, raw_offset, loc_mode, guard_eps=0.0):
        pre, eps = _c(pre), _c(eps)
        loc, scale, sample, _ = H.gauss_sample_fwd(pre, eps, raw_offset, loc_mode, (0., 1., 0., 1.), want_kl=False,
                                                   guard_eps=guard_eps)
        ctx.save_for_backward(pre, eps, loc, scale)
        ctx.cfg = (raw_offset, loc_mode, guard_eps)
        return loc, scale, sample

    @staticmethod
    def backward(ctx, dloc, dscale, dsample):
        pre, eps, loc, scale = ctx.saved_tensors
        raw_offset, loc_mode, guard_eps = ctx.cfg
        # dsample flows to (loc, scale) through the reparameterisation; direct dloc/dscale (KL terms) are folded in by
        # expressing them as an equivalent dsample/eps-free contribution: d pre = J^T [dloc + dsample, dscale + dsample*eps]
        dl = _c(dloc + dsample)
        dsc = _c(dscale + dsample * eps)
        D = loc.shape[1]
        dpre = torch.empty_like(pre)
        if loc_mode == 1:
            idx = torch.arange(D, device=loc.device)
            jac = torch.where(idx % 2 == 1, 1.0 - loc * loc, loc * (1.0 - loc))
            dpre[:, :D] = dl * jac
        else:
            dpre[:, :D] = dl
        dsp = torch.sigmoid(pre[:, D:] + raw_offset)
        if guard_eps > 0:                                   # a floored scale passes no gradient (include/air_hip.h: guard_eps)
            dsp = torch.where(scale <= guard_eps, torch.zeros_like(dsp), dsp)
        dpre[:, D:] = dsc * dsp
        return dpre, None, None, None, None


def gauss_sample(pre, eps, raw_offset=0.0, loc_mode=0, guard_eps=0.0):
    return _GaussSample.apply(pre, eps, float(raw_offset), int(loc_mode), float(guard_eps or 0.0))


class _NormalKL(torch.autograd.Function):
    """sum_d KL(N(loc,scale) || N(prior))   (model.py:174-209)"""

    @staticmethod
    def forward(ctx, loc, scale, prior4):
        loc, scale = _c(loc), _c(scale)
        ctx.save_for_backward(loc, scale)
        ctx.prior4 = prior4
        return H.normal_kl_fwd(loc, scale, prior4)

    @staticmethod
    def backward(ctx, dkl):
        loc, scale = ctx.saved_tensors
        dloc, dscale = H.normal_kl_bwd(loc, scale, ctx.prior4, _c(dkl))
        return dloc, dscale, None


def normal_kl_rows(loc, scale, prior4):
    lead = loc.shape[:-1]
    return _NormalKL.apply(loc.reshape(-1, loc.shape[-1]), scale.reshape(-1, scale.shape[-1]),
                           tuple(float(v) for v in prior4)).reshape(lead)


class _Presence(torch.autograd.Function):
    """presence_prob, presence from the steps-predictor logit (cell.py:137-151); [T,B] time-major"""

    @staticmethod
    def forward(ctx, logit, u, presence_in, step_bias, explore_eps, discrete):
        logit = _c(logit)
        prob, pres = H.presence_fwd(logit, _c(u), step_bias, explore_eps, discrete, _c(presence_in))
        ctx.save_for_backward(logit)
        ctx.cfg = (step_bias, explore_eps, discrete)
        if discrete:
            ctx.mark_non_differentiable(pres)
        return prob, pres

    @staticmethod
    def backward(ctx, dprob, dpres):
        logit, = ctx.saved_tensors
        step_bias, explore_eps, discrete = ctx.cfg
        dlogit = H.presence_bwd(logit, step_bias, explore_eps, discrete, _c(dprob), None if discrete else _c(dpres))
        return dlogit, None, None, None, None, None


def presence(logit, u, presence_in, step_bias, explore_eps, discrete):
    return _Presence.apply(logit, u, presence_in, float(step_bias), explore_eps, bool(discrete))


class _RecLoglik(torch.autograd.Function):
    """per-sample -log N(obs | mult*canvas, std) summed over pixels (model.py:319-324)"""

    @staticmethod
    def forward(ctx, obs, canvas, mult, std):
        obs, canvas = _c(obs), _c(canvas)
        ctx.save_for_backward(obs, canvas)
        ctx.cfg = (mult, std)
        return H.rec_loglik_fwd(obs, canvas, mult, std)

    @staticmethod
    def backward(ctx, dps):
        obs, canvas = ctx.saved_tensors
        return None, H.rec_loglik_bwd(obs, canvas, ctx.cfg[0], ctx.cfg[1], _c(dps)), None, None


def rec_loglik(obs, canvas, mult, std):
    return _RecLoglik.apply(obs, canvas, float(mult), float(std))


class _NumSteps(torch.autograd.Function):
    """q(n), KL(q||prior) per sample, step weights, log q(n_sampled)  (prior.py:62-151, model.py:139-163); fp64 inside"""

    @staticmethod
    def forward(ctx, presence_prob, presence, prior_f64):
        presence_prob, presence = _c(presence_prob), _c(presence)
        q, kl, logp, w = H.numsteps_fwd(presence_prob, presence, prior_f64)
        ctx.save_for_backward(presence_prob, presence, prior_f64)
        ctx.mark_non_differentiable(q)
        return q, kl, logp, w

    @staticmethod
    def backward(ctx, dq, dkl, dlogp, dw):
        presence_prob, presence, prior = ctx.saved_tensors
        # kernel contract: kl_scale is a scalar; per-sample dkl is folded in by linearity (one call per distinct term)
        B = presence_prob.shape[1]
        if dkl is not None and not bool((dkl == dkl.reshape(-1)[0]).all()):
            raise NotImplementedError("numsteps backward expects a uniform weight on kl_per_sample (mean/sum)")
        scale = float(dkl.reshape(-1)[0]) if dkl is not None else 0.0
        dprob = H.numsteps_bwd(presence_prob, presence, prior, scale, _c(dw), _c(dlogp))
        return dprob, None, None


def numsteps(presence_prob, presence, prior_f64):
    return _NumSteps.apply(presence_prob, presence, prior_f64)
