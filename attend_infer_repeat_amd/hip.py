"""Tensor-level entry points over the C ABI (include/air_hip.h): argument checking, output allocation and stream
plumbing only -- all arithmetic happens in libair_hip.so.  No autograd here (see functional.py) and no fallback:
CPU tensors are rejected.

torch is used for device memory and the current HIP stream, nothing else.
"""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_ELU = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_ELU, EPI_MUL_DELU, EPI_ADD_AUX, EPI_ADD_AUX_ELU = 0, 1, 2, 3, 4, 5

_WS = {}
_WS_BYTES = 64 << 20


def lib():
    return _lib.load()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name, dims=None):
    if t is None:
        return None
    if not torch.is_tensor(t) or not t.is_cuda:
        raise _lib.AirHipError(f"{name}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.AirHipError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.AirHipError(f"{name}: expected a contiguous tensor")
    if dims is not None and t.dim() != dims:
        raise _lib.AirHipError(f"{name}: expected {dims} dims, got shape {tuple(t.shape)}")
    return t


def workspace(device=None):
    """Split-K workspace shared by every GEMM on a device (allocated once, outside any graph capture)."""
    dev = torch.device(device if device is not None else torch.cuda.current_device())
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _WS:
        _WS[key] = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", key))
    return _WS[key]


def _ws_args(t):
    ws = workspace(t.device)
    return _p(ws), ctypes.c_size_t(ws.numel() * 4)


# ---- spatial transformer ----------------------------------------------------------------------------------------
def st_read_fwd(img, where, crop_size, n_img=None):
    img = _f32(img, "img", 3); where = _f32(where, "where", 2)
    n_images, H, W = img.shape
    n = where.shape[0]
    h, w = int(crop_size[0]), int(crop_size[1])
    out = torch.empty((n, h, w), dtype=torch.float32, device=img.device)
    _lib.check(lib().air_st_read_fwd(_p(img), _p(where), _p(out), n, n_images if n_img is None else n_img, H, W, h,
                                     w, _stream()), "air_st_read_fwd")
    return out


def st_read_bwd(img, where, dglimpse, want_dimg=False):
    img = _f32(img, "img", 3); where = _f32(where, "where", 2); dglimpse = _f32(dglimpse, "dglimpse", 3)
    n_images, H, W = img.shape
    n, h, w = dglimpse.shape
    dwhere = torch.empty((n, 4), dtype=torch.float32, device=img.device)
    dimg = torch.empty_like(img) if want_dimg else None
    _lib.check(lib().air_st_read_bwd(_p(img), _p(where), _p(dglimpse), _p(dwhere), _p(dimg), n, n_images, H, W, h, w,
                                     _stream()), "air_st_read_bwd")
    return dwhere, dimg


def st_write_fwd(glimpse, where, img_size, presence=None, canvas_in=None):
    glimpse = _f32(glimpse, "glimpse", 3); where = _f32(where, "where", 2)
    presence = _f32(presence, "presence"); canvas_in = _f32(canvas_in, "canvas_in")
    n, h, w = glimpse.shape
    H, W = int(img_size[0]), int(img_size[1])
    out = torch.empty((n, H, W), dtype=torch.float32, device=glimpse.device)
    _lib.check(lib().air_st_write_fwd(_p(glimpse), _p(where), _p(presence), _p(canvas_in), _p(out), n, H, W, h, w,
                                      _stream()), "air_st_write_fwd")
    return out


def st_write_bwd(glimpse, where, dcanvas, presence=None, want_dpresence=False):
    glimpse = _f32(glimpse, "glimpse", 3); where = _f32(where, "where", 2); dcanvas = _f32(dcanvas, "dcanvas", 3)
    presence = _f32(presence, "presence")
    n, h, w = glimpse.shape
    H, W = dcanvas.shape[1:]
    dg = torch.empty_like(glimpse)
    dwhere = torch.empty((n, 4), dtype=torch.float32, device=glimpse.device)
    dpres = torch.empty((n,), dtype=torch.float32, device=glimpse.device) if want_dpresence else None
    _lib.check(lib().air_st_write_bwd(_p(glimpse), _p(where), _p(presence), _p(dcanvas), _p(dg), _p(dwhere),
                                      _p(dpres), n, H, W, h, w, _stream()), "air_st_write_bwd")
    return dg, dwhere, dpres


def canvas_unroll_fwd(glimpse, where, presence, img_size, obs=None, mult=1.0, std=1.0, keep_steps=True):
    """glimpse[T,B,h,w], where[T,B,4], presence[T,B] -> (canvas_steps[T,B,H,W] | None, final[B,H,W], rec[B] | None)"""
    glimpse = _f32(glimpse, "glimpse", 4); where = _f32(where, "where", 3); presence = _f32(presence, "presence")
    obs = _f32(obs, "obs")
    T, B, h, w = glimpse.shape
    H, W = int(img_size[0]), int(img_size[1])
    dev = glimpse.device
    steps = torch.empty((T, B, H, W), dtype=torch.float32, device=dev) if keep_steps else None
    final = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    rec = torch.empty((B,), dtype=torch.float32, device=dev) if obs is not None else None
    _lib.check(lib().air_canvas_unroll_fwd(_p(glimpse), _p(where), _p(presence), _p(obs), _p(steps), _p(final),
                                           _p(rec), T, B, H, W, h, w, float(mult), float(std), _stream()),
               "air_canvas_unroll_fwd")
    return steps, final, rec


def canvas_unroll_bwd(glimpse, where, presence, obs, final_canvas, mult, std, loss_scale):
    """final_canvas=None: the recompute form (every (t, b) unit re-forms the canvas on its own footprint from the T glimpses of
    its image, bit-identically to the forward) -- the backward then does not depend on the forward launch."""
    glimpse = _f32(glimpse, "glimpse", 4); where = _f32(where, "where", 3); presence = _f32(presence, "presence")
    obs = _f32(obs, "obs", 3)
    final_canvas = _f32(final_canvas, "final_canvas", 3) if final_canvas is not None else None
    T, B, h, w = glimpse.shape
    H, W = obs.shape[1:]
    dg = torch.empty_like(glimpse)
    dwhere = torch.empty((T, B, 4), dtype=torch.float32, device=glimpse.device)
    _lib.check(lib().air_canvas_unroll_bwd(_p(glimpse), _p(where), _p(presence), _p(obs), _p(final_canvas), _p(dg),
                                           _p(dwhere), T, B, H, W, h, w, float(mult), float(std), float(loss_scale),
                                           _stream()), "air_canvas_unroll_bwd")
    return dg, dwhere


# ---- dense ---------------------------------------------------------------------------------------------------------
def gemm(A, B, ta=False, tb=False, bias=None, epilogue=EPI_NONE, aux=None, beta=0.0, out=None, colsum=False,
         use_workspace=True):
    """C = epi(op(A).op(B) + beta*C).  A, B: 2-D tensors whose last stride is 1 (row views with a leading dimension
    are fine).  Returns C (and the column sums of op(B) when colsum=True)."""
    for t, nm in ((A, "A"), (B, "B")):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise _lib.AirHipError(f"gemm: {nm} must be a 2-D float32 CUDA tensor with unit inner stride")
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    Kb = B.shape[1] if tb else B.shape[0]
    N = B.shape[0] if tb else B.shape[1]
    if K != Kb:
        raise _lib.AirHipError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    lda = A.stride(0) if A.shape[0] > 1 else A.shape[1]
    ldb = B.stride(0) if B.shape[0] > 1 else B.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    ldc = out.stride(0) if out.shape[0] > 1 else out.shape[1]
    ldaux = 0
    if aux is not None:
        ldaux = aux.stride(0) if aux.shape[0] > 1 else aux.shape[1]
    cs = torch.empty((N,), dtype=torch.float32, device=A.device) if colsum else None
    wsp, wsb = _ws_args(A) if use_workspace else (None, ctypes.c_size_t(0))
    _lib.check(lib().air_gemm(int(ta), int(tb), M, N, K, _p(A), lda, _p(B), ldb, _p(out), ldc, _p(bias),
                              int(epilogue), _p(aux), ldaux, float(beta), _p(cs), wsp, wsb, _stream()), "air_gemm")
    return (out, cs) if colsum else out


def gemm_grouped(problems, precision=0):
    """Several independent GEMMs in one launch (air_gemm_grouped).  problems: dicts with A, B and optional ta, tb, bias,
    epilogue, aux, beta, out, colsum (bool); a single problem may carry the K-split consumer prologue A2, a_bias, a_elu, a_out
    (a = act(A + A2 + a_bias), see AirGemmDesc).  bf16 data path (precision=1): A16 / B16 = bf16 mirrors of A / B (same shape
    and strides in elements), C16 = bf16 tensor that receives bf16(C).  Returns [(C, colsum|None)]."""
    descs, outs, keep = [], [], []
    for pr in problems:
        A, B = pr["A"], pr["B"]
        ta, tb = bool(pr.get("ta", False)), bool(pr.get("tb", False))
        M = A.shape[1] if ta else A.shape[0]
        K = A.shape[0] if ta else A.shape[1]
        N = B.shape[0] if tb else B.shape[1]
        out = pr.get("out")
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        aux, bias = pr.get("aux"), pr.get("bias")
        cs = torch.empty((N,), dtype=torch.float32, device=A.device) if pr.get("colsum") else None
        ld = lambda t: t.stride(0) if t.shape[0] > 1 else t.shape[1]
        d = _lib.AirGemmDesc(int(ta), int(tb), M, N, K, A.data_ptr(), ld(A), B.data_ptr(), ld(B), out.data_ptr(), ld(out),
                             bias.data_ptr() if bias is not None else None, int(pr.get("epilogue", EPI_NONE)),
                             aux.data_ptr() if aux is not None else None, ld(aux) if aux is not None else 0,
                             float(pr.get("beta", 0.0)), cs.data_ptr() if cs is not None else None, int(precision),
                             pr["A2"].data_ptr() if pr.get("A2") is not None else None,
                             pr["a_bias"].data_ptr() if pr.get("a_bias") is not None else None, int(bool(pr.get("a_elu", False))),
                             pr["a_out"].data_ptr() if pr.get("a_out") is not None else None)
        for fld in ("A16", "B16", "C16"):
            t = pr.get(fld)
            if t is not None:
                if t.dtype != torch.bfloat16 or not t.is_cuda:
                    raise _lib.AirHipError(f"gemm_grouped: {fld} must be a bfloat16 CUDA tensor")
                setattr(d, fld, t.data_ptr())
        descs.append(d); outs.append((out, cs)); keep.append((A, B, aux, bias, pr.get("A16"), pr.get("B16"), pr.get("C16")))
    arr = (_lib.AirGemmDesc * len(descs))(*descs)
    _lib.check(lib().air_gemm_grouped(arr, len(descs), _stream()), "air_gemm_grouped")
    return outs


def linear_fwd(x, w, b, act):
    x = _f32(x, "x", 2); w = _f32(w, "w", 2); b = _f32(b, "b")
    M, K = x.shape
    N = w.shape[1]
    if w.shape[0] != K:
        raise _lib.AirHipError(f"linear: x is [{M},{K}] but w is {tuple(w.shape)}")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    wsp, wsb = _ws_args(x)
    _lib.check(lib().air_linear_fwd(_p(x), _p(w), _p(b), _p(y), M, K, N, int(act), wsp, wsb, _stream()),
               "air_linear_fwd")
    return y


def linear_bwd(x, w, y, dy, act, want_dx=True, want_db=True):
    x = _f32(x, "x", 2); w = _f32(w, "w", 2); dy = _f32(dy, "dy", 2); y = _f32(y, "y")
    M, K = x.shape
    N = w.shape[1]
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.empty_like(w)
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if want_db else None
    gbuf = torch.empty_like(dy) if act != ACT_NONE else None
    wsp, wsb = _ws_args(x)
    _lib.check(lib().air_linear_bwd(_p(x), _p(w), _p(y), _p(dy), _p(dx), _p(dw), _p(db), _p(gbuf), M, K, N, int(act),
                                    wsp, wsb, _stream()), "air_linear_bwd")
    return dx, dw, db


def lstm_pointwise_fwd(gates, c_prev, forget_bias=1.0):
    gates = _f32(gates, "gates", 2); c_prev = _f32(c_prev, "c_prev", 2)
    M, Hd = c_prev.shape
    h = torch.empty_like(c_prev); c = torch.empty_like(c_prev); act = torch.empty_like(gates)
    _lib.check(lib().air_lstm_pointwise_fwd(_p(gates), _p(c_prev), _p(h), _p(c), _p(act), M, Hd, float(forget_bias),
                                            _stream()), "air_lstm_pointwise_fwd")
    return h, c, act


def lstm_pointwise_bwd(gate_act, c_prev, c, dh, dc):
    gate_act = _f32(gate_act, "gate_act", 2); c_prev = _f32(c_prev, "c_prev", 2); c = _f32(c, "c", 2)
    dh = _f32(dh, "dh"); dc = _f32(dc, "dc")
    M, Hd = c_prev.shape
    dgates = torch.empty_like(gate_act); dc_prev = torch.empty_like(c_prev)
    _lib.check(lib().air_lstm_pointwise_bwd(_p(gate_act), _p(c_prev), _p(c), _p(dh), None, _p(dc), _p(dgates),
                                            _p(dc_prev), M, Hd, _stream()), "air_lstm_pointwise_bwd")
    return dgates, dc_prev


def lstm_step_fwd(h_prev, c_prev, w_h, gx, forget_bias=1.0, precision=0):
    """fused recurrent product + gate math: (h, c, gate_act) from h_prev[M,Hd], c_prev[M,Hd], w_h[Hd,4Hd], gx[M,4Hd]"""
    h_prev = _f32(h_prev, "h_prev", 2); c_prev = _f32(c_prev, "c_prev", 2); gx = _f32(gx, "gx", 2)
    if not (w_h.is_cuda and w_h.dtype == torch.float32 and w_h.dim() == 2 and w_h.stride(1) == 1):
        raise ValueError("w_h must be a row-major fp32 CUDA matrix (row stride allowed)")
    M, Hd = c_prev.shape
    h = torch.empty_like(c_prev); c = torch.empty_like(c_prev); act = torch.empty_like(gx)
    _lib.check(lib().air_lstm_step_fwd(_p(h_prev), _p(c_prev), _p(w_h), w_h.stride(0), _p(gx), gx.stride(0), _p(h), _p(c),
                                       _p(act), M, Hd, float(forget_bias), int(precision), _stream()), "air_lstm_step_fwd")
    return h, c, act


def lstm_step_bwd(dgates_next, w_h, dh_a, dh_b, dc_in, gate_act, c_prev, c, dgx_in=None, want_dgx=False, precision=0):
    """fused BPTT link: dh = dgates_next . w_h^T + dh_a + dh_b, then the pointwise backward -> dgates, dc_prev[, dgx]"""
    dgates_next = _f32(dgates_next, "dgates_next", 2); w_h = _f32(w_h, "w_h", 2)
    gate_act = _f32(gate_act, "gate_act", 2); c_prev = _f32(c_prev, "c_prev", 2); c = _f32(c, "c", 2)
    M, Hd = c_prev.shape
    dgates = torch.empty_like(gate_act); dc_prev = torch.empty_like(c_prev)
    dgx = torch.empty_like(gate_act) if want_dgx else None
    _lib.check(lib().air_lstm_step_bwd(_p(dgates_next), _p(w_h), _p(dh_a), _p(dh_b), _p(dc_in), _p(gate_act), _p(c_prev),
                                       _p(c), _p(dgx_in), _p(dgates), _p(dc_prev), _p(dgx), M, Hd, int(precision),
                                       _stream()), "air_lstm_step_bwd")
    return dgates, dc_prev, dgx


def rmsprop_slice(p, g, ms, mg, mom, lo, hi, n_model, lr_dev, lr_mult_tail=1.0, decay=0.9, momentum=0.9, eps=1e-10,
                  grad_scale=1.0):
    """AirRmspropSlice over the flat buffers: elements [lo, hi) are updated by the *_opt launch the slice is handed to."""
    for t, nm in ((p, "p"), (g, "g"), (ms, "ms"), (mg, "mg"), (mom, "mom"), (lr_dev, "lr_dev")):
        _f32(t, nm)
    return _lib.AirRmspropSlice(p.data_ptr(), g.data_ptr(), ms.data_ptr(), mg.data_ptr(), mom.data_ptr(), int(lo), int(hi),
                                int(n_model), lr_dev.data_ptr(), float(lr_mult_tail), float(decay), float(momentum), float(eps),
                                float(grad_scale))


def lstm_pointwise_bwd_opt(gate_act, c_prev, c, dh, dc, opt):
    """air_lstm_pointwise_bwd with an optimiser slice (rmsprop_slice(...)) riding as extra workgroups of the launch"""
    gate_act = _f32(gate_act, "gate_act", 2); c_prev = _f32(c_prev, "c_prev", 2); c = _f32(c, "c", 2)
    dh = _f32(dh, "dh"); dc = _f32(dc, "dc")
    M, Hd = c_prev.shape
    dgates = torch.empty_like(gate_act); dc_prev = torch.empty_like(c_prev)
    _lib.check(lib().air_lstm_pointwise_bwd_opt(_p(gate_act), _p(c_prev), _p(c), _p(dh), None, _p(dc), _p(dgates),
                                                _p(dc_prev), M, Hd, ctypes.byref(opt) if opt is not None else None, _stream()),
               "air_lstm_pointwise_bwd_opt")
    return dgates, dc_prev


def lstm_step_bwd_opt(dgates_next, w_h, dh_a, dh_b, dc_in, gate_act, c_prev, c, opt, precision=0):
    """air_lstm_step_bwd with an optimiser slice riding along"""
    dgates_next = _f32(dgates_next, "dgates_next", 2); w_h = _f32(w_h, "w_h", 2)
    gate_act = _f32(gate_act, "gate_act", 2); c_prev = _f32(c_prev, "c_prev", 2); c = _f32(c, "c", 2)
    M, Hd = c_prev.shape
    dgates = torch.empty_like(gate_act); dc_prev = torch.empty_like(c_prev)
    _lib.check(lib().air_lstm_step_bwd_opt(_p(dgates_next), _p(w_h), _p(dh_a), _p(dh_b), _p(dc_in), _p(gate_act), _p(c_prev),
                                           _p(c), None, _p(dgates), _p(dc_prev), None, M, Hd, int(precision),
                                           ctypes.byref(opt) if opt is not None else None, _stream()), "air_lstm_step_bwd_opt")
    return dgates, dc_prev


def lstm_step_bwd_entry(gate_act1, c_prev1, c1, dh_a1, dh_b1, w_h, dh_a, dh_b, gate_act, c_prev, c, opt=None, want_dgx=True):
    """air_lstm_step_bwd_entry: pointwise backward of the last step + the first BPTT link in one launch.
    Returns (dgates1, dc_prev1, dgates, dc_prev, dgx)"""
    gate_act1 = _f32(gate_act1, "gate_act1", 2); c_prev1 = _f32(c_prev1, "c_prev1", 2); c1 = _f32(c1, "c1", 2)
    w_h = _f32(w_h, "w_h", 2); gate_act = _f32(gate_act, "gate_act", 2); c_prev = _f32(c_prev, "c_prev", 2); c = _f32(c, "c", 2)
    M, Hd = c_prev.shape
    dgates1 = torch.empty_like(gate_act1); dc_prev1 = torch.empty_like(c_prev1)
    dgates = torch.empty_like(gate_act); dc_prev = torch.empty_like(c_prev)
    dgx = torch.empty_like(gate_act) if want_dgx else None
    _lib.check(lib().air_lstm_step_bwd_entry(_p(gate_act1), _p(c_prev1), _p(c1), _p(dh_a1), _p(dh_b1), _p(dgates1), _p(dc_prev1),
                                             _p(w_h), _p(dh_a), _p(dh_b), _p(gate_act), _p(c_prev), _p(c), _p(dgates), _p(dc_prev),
                                             _p(dgx), M, Hd, ctypes.byref(opt) if opt is not None else None, _stream()),
               "air_lstm_step_bwd_entry")
    return dgates1, dc_prev1, dgates, dc_prev, dgx


# ---- stochastic nodes ----------------------------------------------------------------------------------------------
def gauss_sample_fwd(pre, eps, raw_offset, loc_mode, prior4, want_kl=True, guard_eps=0.0):
    """pre[M, >=2D] (row stride allowed), eps[M,D] or None -> loc, scale, sample|None, kl_row|None
    guard_eps > 0: the stability switch of include/air_hip.h (scale floor, |where scale| >= guard_eps); 0 = the reference's arithmetic"""
    if not (pre.is_cuda and pre.dtype == torch.float32 and pre.dim() == 2 and pre.stride(1) == 1):
        raise _lib.AirHipError("gauss_sample: pre must be a 2-D float32 CUDA tensor with unit inner stride")
    eps = _f32(eps, "eps")
    M = pre.shape[0]
    D = pre.shape[1] // 2
    ld = pre.stride(0) if M > 1 else pre.shape[1]
    dev = pre.device
    loc = torch.empty((M, D), dtype=torch.float32, device=dev); scale = torch.empty_like(loc)
    sample = torch.empty_like(loc) if eps is not None else None
    kl = torch.empty((M,), dtype=torch.float32, device=dev) if want_kl else None
    a, b, c, d = (float(v) for v in prior4)
    _lib.check(lib().air_gauss_sample_fwd(_p(pre), ld, _p(eps), float(raw_offset), int(loc_mode), a, b, c, d, _p(loc),
                                          _p(scale), _p(sample), _p(kl), M, D, float(guard_eps), _stream()), "air_gauss_sample_fwd")
    return loc, scale, sample, kl


def gauss_sample_bwd(pre, eps, raw_offset, loc_mode, prior4, loc, scale, dsample, dkl_row, guard_eps=0.0):
    M, D = loc.shape
    ld = pre.stride(0) if M > 1 else pre.shape[1]
    dpre = torch.empty((M, 2 * D), dtype=torch.float32, device=pre.device)
    a, b, c, d = (float(v) for v in prior4)
    _lib.check(lib().air_gauss_sample_bwd(_p(pre), ld, _p(eps), float(raw_offset), int(loc_mode), a, b, c, d, _p(loc),
                                          _p(scale), _p(_f32(dsample, "dsample")), None, _p(_f32(dkl_row, "dkl_row")),
                                          1.0, _p(dpre), 2 * D, M, D, float(guard_eps), None, 0, None, _stream()), "air_gauss_sample_bwd")
    return dpre


def normal_kl_fwd(loc, scale, prior4):
    loc = _f32(loc, "loc", 2); scale = _f32(scale, "scale", 2)
    M, D = loc.shape
    kl = torch.empty((M,), dtype=torch.float32, device=loc.device)
    a, b, c, d = (float(v) for v in prior4)
    _lib.check(lib().air_normal_kl_fwd(_p(loc), _p(scale), a, b, c, d, _p(kl), M, D, _stream()), "air_normal_kl_fwd")
    return kl


def normal_kl_bwd(loc, scale, prior4, dkl_row):
    M, D = loc.shape
    dloc = torch.empty_like(loc); dscale = torch.empty_like(scale)
    a, b, c, d = (float(v) for v in prior4)
    _lib.check(lib().air_normal_kl_bwd(_p(loc), _p(scale), a, b, c, d, _p(_f32(dkl_row, "dkl_row", 1)), _p(dloc),
                                       _p(dscale), M, D, _stream()), "air_normal_kl_bwd")
    return dloc, dscale


def presence_fwd(logit, u, step_bias, explore_eps, discrete, presence_in=None):
    """logit, u: [T,B] -> presence_prob[T,B], presence[T,B]"""
    logit = _f32(logit, "logit", 2); u = _f32(u, "u"); presence_in = _f32(presence_in, "presence_in")
    T, B = logit.shape
    prob = torch.empty_like(logit); pres = torch.empty_like(logit)
    eps = -1.0 if explore_eps is None else float(explore_eps)
    _lib.check(lib().air_presence_fwd(_p(logit), _p(u), _p(presence_in), float(step_bias), eps, int(bool(discrete)),
                                      _p(prob), _p(pres), T, B, _stream()), "air_presence_fwd")
    return prob, pres


def presence_bwd(logit, step_bias, explore_eps, discrete, dprob, dpres=None):
    logit = _f32(logit, "logit", 2)
    T, B = logit.shape
    dlogit = torch.empty_like(logit)
    eps = -1.0 if explore_eps is None else float(explore_eps)
    _lib.check(lib().air_presence_bwd(_p(logit), float(step_bias), eps, int(bool(discrete)), _p(_f32(dprob, "dprob")),
                                      _p(_f32(dpres, "dpres")), _p(dlogit), T, B, _stream()), "air_presence_bwd")
    return dlogit


# ---- objective -----------------------------------------------------------------------------------------------------
def rec_loglik_fwd(obs, canvas, mult, std):
    obs = _f32(obs, "obs"); canvas = _f32(canvas, "canvas")
    B = obs.shape[0]
    P = obs.numel() // B
    out = torch.empty((B,), dtype=torch.float32, device=obs.device)
    _lib.check(lib().air_rec_loglik_fwd(_p(obs), _p(canvas), float(mult), float(std), _p(out), B, P, _stream()),
               "air_rec_loglik_fwd")
    return out


def rec_loglik_bwd(obs, canvas, mult, std, dper_sample=None, scale=1.0):
    obs = _f32(obs, "obs"); canvas = _f32(canvas, "canvas")
    B = obs.shape[0]
    P = obs.numel() // B
    dc = torch.empty_like(canvas)
    _lib.check(lib().air_rec_loglik_bwd(_p(obs), _p(canvas), float(mult), float(std),
                                        _p(_f32(dper_sample, "dper_sample")), float(scale), _p(dc), B, P, _stream()),
               "air_rec_loglik_bwd")
    return dc


def numsteps_fwd(presence_prob, presence, prior_f64):
    presence_prob = _f32(presence_prob, "presence_prob", 2); presence = _f32(presence, "presence")
    if prior_f64.dtype != torch.float64 or not prior_f64.is_cuda:
        raise _lib.AirHipError("numsteps: prior must be a float64 CUDA tensor")
    T, B = presence_prob.shape
    dev = presence_prob.device
    q = torch.empty((B, T + 1), dtype=torch.float32, device=dev)
    kl = torch.empty((B,), dtype=torch.float32, device=dev)
    logp = torch.empty((B,), dtype=torch.float32, device=dev) if presence is not None else None
    w = torch.empty((T, B), dtype=torch.float32, device=dev)
    _lib.check(lib().air_numsteps_fwd(_p(presence_prob), _p(presence), _p(prior_f64), _p(q), _p(kl), _p(logp), _p(w),
                                      T, B, _stream()), "air_numsteps_fwd")
    return q, kl, logp, w


def numsteps_bwd(presence_prob, presence, prior_f64, kl_scale, dstep_weight=None, dlogp=None):
    presence_prob = _f32(presence_prob, "presence_prob", 2)
    T, B = presence_prob.shape
    dprob = torch.empty_like(presence_prob)
    _lib.check(lib().air_numsteps_bwd(_p(presence_prob), _p(_f32(presence, "presence")), _p(prior_f64),
                                      float(kl_scale), _p(_f32(dstep_weight, "dstep_weight")),
                                      _p(_f32(dlogp, "dlogp")), _p(dprob), T, B, _stream()), "air_numsteps_bwd")
    return dprob


def nvil(imp, baseline, logp, ema=None):
    imp = _f32(imp, "imp", 1); baseline = _f32(baseline, "baseline", 1); logp = _f32(logp, "logp", 1)
    B = imp.shape[0]
    dev = imp.device
    out = torch.empty((4,), dtype=torch.float32, device=dev)
    dlogp = torch.empty((B,), dtype=torch.float32, device=dev); dbase = torch.empty((B,), dtype=torch.float32, device=dev)
    _lib.check(lib().air_nvil(_p(imp), _p(baseline), _p(logp), _p(out), _p(dlogp), _p(dbase), B, _p(ema), _stream()), "air_nvil")
    return out, dlogp, dbase


def baseline_pack(img, what, where, presence, state_parts):
    """img[B,...], what[T,B,A], where[T,B,4], presence[T,B(,1)], state_parts: list of up to two [B,S] tensors."""
    img = _f32(img, "img"); what = _f32(what, "what", 3); where = _f32(where, "where", 3)
    presence = _f32(presence, "presence")
    T, B, A = what.shape
    Pn = img.numel() // B
    parts = [_f32(s, "state", 2) for s in state_parts]
    if len(parts) > 2:
        raise _lib.AirHipError("baseline_pack: at most two state parts")
    s0 = parts[0] if len(parts) > 0 else None
    s1 = parts[1] if len(parts) > 1 else None
    S0 = s0.shape[1] if s0 is not None else 0
    S1 = s1.shape[1] if s1 is not None else 0
    out = torch.empty((B, Pn + T * A + T * 4 + T + S0 + S1), dtype=torch.float32, device=img.device)
    _lib.check(lib().air_baseline_pack(_p(img), _p(what), _p(where), _p(presence), _p(s0), _p(s1), _p(out), T, B, Pn,
                                       A, S0, S1, _stream()), "air_baseline_pack")
    return out


# ---- optimiser / noise / utils -------------------------------------------------------------------------------------
def rmsprop_centered_(p, g, ms, mg, mom, lr_dev, lr_mult=1.0, decay=0.9, momentum=0.9, eps=1e-10, grad_scale=1.0,
                      centered=True):
    """tf.train.RMSPropOptimizer's update (all of its keywords) in place; `centered=True, momentum=.9` is the reference's
    choice (model.py:265)."""
    for t, nm in ((p, "p"), (g, "g"), (ms, "ms"), (mg, "mg"), (mom, "mom"), (lr_dev, "lr_dev")):
        _f32(t, nm)
    _lib.check(lib().air_rmsprop(_p(p), _p(g), _p(ms), _p(mg), _p(mom), ctypes.c_size_t(p.numel()), _p(lr_dev),
                                 float(lr_mult), float(decay), float(momentum), float(eps), int(bool(centered)),
                                 float(grad_scale), _stream()), "air_rmsprop")


def rng_fill(state_dev, normal=None, uniform=None, advance=True):
    """state_dev: int64/uint64 CUDA tensor [2] = {seed, offset}.  Fills the given flat tensors in place."""
    nn = normal.numel() if normal is not None else 0
    nu = uniform.numel() if uniform is not None else 0
    _lib.check(lib().air_rng_fill(_p(normal), ctypes.c_size_t(nn), _p(uniform), ctypes.c_size_t(nu), _p(state_dev),
                                  _stream()), "air_rng_fill")
    if advance:
        _lib.check(lib().air_rng_advance(_p(state_dev), ctypes.c_uint64((nn + 3) // 4 + (nu + 3) // 4), _stream()),
                   "air_rng_advance")


def fill_(t, v):
    _lib.check(lib().air_fill(_p(t), ctypes.c_size_t(t.numel()), float(v), _stream()), "air_fill")
    return t


def axpby(a, alpha, b=None, beta=0.0, out=None):
    out = torch.empty_like(a) if out is None else out
    _lib.check(lib().air_axpby(_p(a), float(alpha), _p(b), float(beta), _p(out), ctypes.c_size_t(a.numel()),
                               _stream()), "air_axpby")
    return out


def tile_rows(src, rows):
    src = _f32(src, "src")
    cols = src.numel()
    out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    _lib.check(lib().air_tile_rows(_p(src), _p(out), rows, cols, _stream()), "air_tile_rows")
    return out


def colsum(x):
    M, N = x.shape
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    _lib.check(lib().air_colsum(_p(x), x.stride(0) if M > 1 else N, _p(out), M, N, _stream()), "air_colsum")
    return out
