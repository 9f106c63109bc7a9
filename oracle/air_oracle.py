"""CPU restatement of the AIR hot path (torch-CPU, fp32 or fp64) -- TEST INFRASTRUCTURE ONLY.

This file restates, from the reference sources, the arithmetic of one AIR train
step: the T-step cell unroll, the ELBO / NVIL objective and the centred-RMSProp
update.  It exists to *check* the HIP path (tests/, __graft_entry__.smoke()) and
to be *timed* as the CPU baseline (bench.py `cpu_baseline`, kind="port").  The
product package never imports it.

PARITY STATUS
  * pinned     : the number-of-steps prior math (geometric_prior, tabular_kl,
                 bernoulli_to_modified_geometric) -- checked against every
                 known answer in the reference's test/prior_test.py
                 (tests/test_oracle_prior.py).
  * UNPINNED   : everything else (ST read/write, LSTM, Gaussian sampling, KL,
                 NVIL, RMSProp).  The reference cannot be imported here (Python 2
                 + TF 1.1 + Sonnet 1.1, none installable) and its own tests hold
                 no numbers for these ops (test/cell_test.py only prints shapes).
                 Semantics of the un-vendored deps (Sonnet v1.1 @3fd7d9d, TF
                 1.1.0rc1) are restated from their public behaviour; the
                 assumptions are listed in tests/golden/ASSUMPTIONS.md.  The ST is
                 cross-checked by two further independent codings
                 (oracle/st_loops.c scalar loops, torch grid_sample) and fp64
                 finite differences; affine/ELU, the LSTM step, Gaussian
                 sampling + KL, the num-steps math and centred RMSProp by a
                 scalar-loop C coding (oracle/net_loops.c,
                 tests/test_oracle_net_loops.py).  Independent codings catch
                 slips, they do not pin the third-party semantics.

Reference citations are `file:line` under /root/reference/attend_infer_repeat/.
Weights use Sonnet layout: Linear w[in, out]; LSTM w_gates[in+hid, 4*hid], gate
order i, j, f, o, forget bias 1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration (scripts/multi_mnist.py:24-94, mnist_model.py:13-44)
# --------------------------------------------------------------------------------------
@dataclass
class AIRConfig:
    img_size: Tuple[int, int] = (50, 50)
    crop_size: Tuple[int, int] = (20, 20)
    n_appearance: int = 50                      # mnist_model.py:34
    n_hidden: int = 256                         # snt.LSTM(256), mnist_model.py:35
    transition: str = "lstm"                    # "lstm" | "gru" (test/cell_test.py:12)
    inpt_encoder_hidden: Sequence[int] = (256, 256)
    glimpse_encoder_hidden: Sequence[int] = (256, 256)
    glimpse_decoder_hidden: Sequence[int] = (256, 256)
    transform_estimator_hidden: Sequence[int] = (256, 256)
    steps_pred_hidden: Sequence[int] = (128, 64)
    baseline_hidden: Sequence[int] = (256, 128)
    max_steps: int = 3
    transform_var_bias: float = 0.5             # multi_mnist.py:56
    step_bias: float = 0.75                     # multi_mnist.py:55
    output_multiplier: float = 0.5              # multi_mnist.py:57
    output_std: float = 0.3                     # mnist_model.py:42
    explore_eps: Optional[float] = 1e-3         # multi_mnist.py:59
    what_scale_offset: float = 0.5              # cell.py:66
    discrete_steps: bool = True
    # priors (multi_mnist.py:40-51)
    what_prior: Tuple[float, float] = (0.0, 1.0)
    where_scale_prior: Tuple[float, float] = (0.0, 1.0)
    where_shift_prior: Tuple[float, float] = (0.0, 1.0)
    nsp_anneal: Optional[str] = "exp"
    nsp_init: float = 1.0 - 1e-15
    nsp_final: float = 1e-7
    nsp_steps_div: float = 1e4
    nsp_steps: float = 1e5
    nsp_hold_init: float = 1e3
    nsp_analytic: bool = True
    nsp_weight: float = 1.0                     # num_steps_prior.weight (model.py:339-340: prior_loss.add(kl, weight=...))
    use_prior: bool = True
    use_reinforce: bool = True
    decay_rate: Optional[float] = None          # model.py:232-239 (None in the script)
    l2_weight: float = 0.0                      # model.py:346-353 (0 in the script)
    # NOT in the reference (default off): the product's documented stability switch (include/air_hip.h `guard_eps`) restated, so
    # that the guarded arithmetic has an oracle too: scale = max(softplus(..), g) with no gradient through a floored scale, and the
    # sampled scale components of `where` kept at |s| >= g (sign kept, straight-through)
    guard_eps: float = 0.0
    # optimiser (model.py:265, multi_mnist.py:24)
    learning_rate: float = 1e-4
    baseline_lr_mult: float = 10.0              # model.py:363
    rms_decay: float = 0.9
    rms_momentum: float = 0.9
    rms_eps: float = 1e-10
    rms_centered: bool = True                   # opt_kwargs of model.py:265 (tf.train.RMSPropOptimizer's own default: False)

    @property
    def n_pix(self) -> int:
        return int(self.img_size[0] * self.img_size[1])

    @property
    def n_crop(self) -> int:
        return int(self.crop_size[0] * self.crop_size[1])

    @property
    def state_width(self) -> int:
        """flattened width of the transition state fed to the baseline (modules.py:135-136)"""
        return 2 * self.n_hidden if self.transition == "lstm" else self.n_hidden

    @property
    def baseline_in(self) -> int:
        T = self.max_steps
        return self.n_pix + T * self.n_appearance + T * 4 + T + self.state_width


def tiny_config(**kw) -> AIRConfig:
    """The configuration of test/cell_test.py:9-30 (img 3x3, crop 2x2, hidden 5/7/11/13/17)."""
    base = dict(img_size=(3, 3), crop_size=(2, 2), n_appearance=10, n_hidden=3,
                inpt_encoder_hidden=(5,), glimpse_encoder_hidden=(7,), glimpse_decoder_hidden=(11,),
                transform_estimator_hidden=(13,), steps_pred_hidden=(17,), baseline_hidden=(6, 4),
                max_steps=3, transform_var_bias=-2.0, step_bias=0.0, explore_eps=None,
                output_multiplier=1.0, output_std=1.0)
    base.update(kw)
    return AIRConfig(**base)


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def mlp_shapes(n_in: int, hiddens: Sequence[int], n_out: Optional[int]) -> List[Tuple[int, int]]:
    sizes = list(hiddens) + ([n_out] if n_out is not None else [])
    shapes, prev = [], n_in
    for s in sizes:
        shapes.append((prev, int(s)))
        prev = int(s)
    return shapes


def param_shapes(cfg: AIRConfig) -> "Dict[str, Tuple[int, ...]]":
    """Canonical, ordered name -> shape map of every trainable tensor (model first, baseline last)."""
    Hd, A = cfg.n_hidden, cfg.n_appearance
    out: Dict[str, Tuple[int, ...]] = {}

    def add_mlp(prefix, n_in, hiddens, n_out):
        for i, (a, b) in enumerate(mlp_shapes(n_in, hiddens, n_out)):
            out[f"{prefix}/{i}/w"] = (a, b)
            out[f"{prefix}/{i}/b"] = (b,)

    add_mlp("input_encoder", cfg.n_pix, cfg.inpt_encoder_hidden, None)          # modules.py:66-76
    enc_out = int(cfg.inpt_encoder_hidden[-1])
    if cfg.transition == "lstm":                                                # Sonnet LSTM: one fused gate matrix
        out["lstm/w_gates"] = (enc_out + Hd, 4 * Hd)
        out["lstm/b_gates"] = (4 * Hd,)
        out["lstm/h0"] = (1, Hd)                                                # trainable initial state, cell.py:103
        out["lstm/c0"] = (1, Hd)
    else:                                                                       # Sonnet GRU (tests only)
        for g in "zrh":
            out[f"gru/w{g}"] = (enc_out, Hd)
            out[f"gru/u{g}"] = (Hd, Hd)
            out[f"gru/b{g}"] = (Hd,)
        out["gru/h0"] = (1, Hd)
    add_mlp("transform", Hd, cfg.transform_estimator_hidden, 8)                 # modules.py:58-63
    add_mlp("steps", Hd, cfg.steps_pred_hidden, 1)                              # modules.py:119-122
    add_mlp("glimpse_encoder", cfg.n_crop, cfg.glimpse_encoder_hidden, None)    # cell.py:153
    out["what/w"] = (int(cfg.glimpse_encoder_hidden[-1]), 2 * A)                # modules.py:20
    out["what/b"] = (2 * A,)
    add_mlp("glimpse_decoder", A, cfg.glimpse_decoder_hidden, cfg.n_crop)       # modules.py:86-91
    add_mlp("baseline", cfg.baseline_in, cfg.baseline_hidden, 1)                # modules.py:125-143
    return out


def is_baseline_param(name: str) -> bool:
    return name.startswith("baseline/")


def _trunc_normal(rng: np.random.Generator, shape, std):
    """TF truncated_normal: resample outside +-2 sigma."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def init_params(cfg: AIRConfig, seed: int = 1, dtype=torch.float32, bias_std: float = 0.0) -> Dict[str, Tensor]:
    """Sonnet default init: w ~ TruncNormal(0, 1/sqrt(fan_in)), b = 0 (neural.py:53 drops the
    custom initialisers).  `bias_std` > 0 randomises biases / initial state so parity tests exercise them."""
    rng = np.random.Generator(np.random.PCG64(seed))
    params = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) == 2 and not name.endswith(("/h0", "/c0")):
            arr = _trunc_normal(rng, shape, 1.0 / math.sqrt(shape[0]))
        else:
            arr = rng.standard_normal(shape) * bias_std if bias_std > 0 else np.zeros(shape)
        params[name] = torch.tensor(arr, dtype=dtype)
    return params


def make_noise(cfg: AIRConfig, batch: int, seed: int = 2, dtype=torch.float32) -> Dict[str, Tensor]:
    rng = np.random.Generator(np.random.PCG64(seed))
    T = cfg.max_steps
    return {
        "eps_where": torch.tensor(rng.standard_normal((T, batch, 4)), dtype=dtype),
        "eps_what": torch.tensor(rng.standard_normal((T, batch, cfg.n_appearance)), dtype=dtype),
        "u_pres": torch.tensor(rng.random((T, batch, 1)), dtype=dtype),
    }


def synthetic_batch(cfg: AIRConfig, batch: int, seed: int = 0, max_objects: int = 2, dtype=torch.float32):
    """Synthetic multi-MNIST-like batch (shapes of data/data.py:35-107): 0..max_objects soft blobs on a
    zero background; `nums` is [max_objects+1, B, 1] one-hot-cumulative (data.py:101-105)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = cfg.img_size
    imgs = np.zeros((batch, H, W), np.float32)
    nums = np.zeros((max_objects + 1, batch, 1), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(batch):
        n = int(rng.integers(0, max_objects + 1))
        nums[:n, b, 0] = 1.0
        for _ in range(n):
            s = max(2, int(min(H, W) * rng.uniform(0.2, 0.45)))
            cy = rng.uniform(s / 2, max(s / 2 + 1e-3, H - s / 2)); cx = rng.uniform(s / 2, max(s / 2 + 1e-3, W - s / 2))
            ang = rng.uniform(0, np.pi); r = s * 0.4
            # anti-aliased stroke: distance to a segment
            x0, y0 = cx - r * np.cos(ang), cy - r * np.sin(ang)
            x1, y1 = cx + r * np.cos(ang), cy + r * np.sin(ang)
            px, py = xx - x0, yy - y0
            dxs, dys = x1 - x0, y1 - y0
            tt = np.clip((px * dxs + py * dys) / (dxs * dxs + dys * dys + 1e-9), 0, 1)
            d = np.sqrt((px - tt * dxs) ** 2 + (py - tt * dys) ** 2)
            imgs[b] = np.maximum(imgs[b], np.clip(1.6 - d, 0, 1))
    return torch.tensor(imgs, dtype=dtype), torch.tensor(nums, dtype=dtype)


# --------------------------------------------------------------------------------------
# layers (neural.py:42-102)
# --------------------------------------------------------------------------------------
_MM_MODE = ["f32"]


class matmul_mode:
    """`with matmul_mode("bf16"):` makes every dense product of the oracle emulate the product's optional bf16-MFMA mode
    (EngineConfig.mfma_dtype="bf16", BASELINE.json configs[4]): both operands rounded to bf16 (round-to-nearest-even),
    exact products, fp32 accumulation -- forward, dX and dW alike.  Not part of the reference (TF1 fp32 only); it exists
    so that the bf16 path is checked against the same arithmetic rather than against a loose tolerance."""

    def __init__(self, mode: str):
        assert mode in ("f32", "bf16")
        self.mode = mode

    def __enter__(self):
        self.prev = _MM_MODE[0]
        _MM_MODE[0] = self.mode

    def __exit__(self, *exc):
        _MM_MODE[0] = self.prev


def _r16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


class _Bf16MatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        xr, wr = _r16(x), _r16(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _r16(dy)
        return dyr @ wr.t(), xr.t() @ dyr


def mm(x: Tensor, w: Tensor) -> Tensor:
    return _Bf16MatMul.apply(x, w) if _MM_MODE[0] == "bf16" else x @ w


def affine(x: Tensor, w: Tensor, b: Tensor, elu: bool) -> Tensor:
    """Affine = transfer(x.W + b), neural.py:56-60."""
    y = mm(x, w) + b
    return F.elu(y) if elu else y


def mlp(x: Tensor, params: Dict[str, Tensor], prefix: str, n_layers: int, last_linear: bool) -> Tensor:
    """MLP: ELU affines, optionally one linear output layer (neural.py:93-102)."""
    for i in range(n_layers):
        is_last = i == n_layers - 1
        x = affine(x, params[f"{prefix}/{i}/w"], params[f"{prefix}/{i}/b"], elu=not (is_last and last_linear))
    return x


def lstm_step(x: Tensor, h: Tensor, c: Tensor, w: Tensor, b: Tensor, forget_bias: float = 1.0):
    """Sonnet v1 LSTM (mnist_model.py:35, cell.py:127): gates=[x,h].W+b; i,j,f,o."""
    g = mm(torch.cat([x, h], -1), w) + b
    i, j, f, o = torch.chunk(g, 4, -1)
    c2 = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    return h2, c2


def gru_step(x: Tensor, h: Tensor, p: Dict[str, Tensor]):
    """Sonnet v1 GRU (used by test/cell_test.py:12 only)."""
    z = torch.sigmoid(x @ p["gru/wz"] + h @ p["gru/uz"] + p["gru/bz"])
    r = torch.sigmoid(x @ p["gru/wr"] + h @ p["gru/ur"] + p["gru/br"])
    a = torch.tanh(x @ p["gru/wh"] + (r * h) @ p["gru/uh"] + p["gru/bh"])
    return (1 - z) * h + z * a


# --------------------------------------------------------------------------------------
# spatial transformer (modules.py:94-109; Sonnet AffineGridWarper + resampler)
# --------------------------------------------------------------------------------------
def linspace_m11(n: int, dtype) -> Tensor:
    """np.linspace(-1, 1, n) evaluated in fp64 then rounded to `dtype` (as a TF constant would be)."""
    v = np.linspace(-1.0, 1.0, n) if n > 1 else np.array([-1.0])
    return torch.tensor(v, dtype=torch.float64).to(dtype)


def resample_bilinear(src: Tensor, x: Tensor, y: Tensor) -> Tensor:
    """snt.resampler semantics: bilinear, zero outside, sample valid iff -1<x<W and -1<y<H.
    src [B,Hs,Ws]; x, y [B,Ho,Wo] in source-pixel units -> [B,Ho,Wo]."""
    B, Hs, Ws = src.shape
    inside = (x > -1) & (y > -1) & (x < Ws) & (y < Hs)
    # Sample points outside the source (incl. NaN / inf coordinates: an inverse warp with sx == 0 has 1/sx = inf) produce 0 and
    # receive EXACTLY zero gradient -- the resampler's registered gradient op skips them (ASSUMPTIONS.md #3).  Their coordinates
    # are therefore replaced by a constant before any arithmetic: a mask applied afterwards would turn 0 * NaN into NaN in the
    # gradient of the OTHER coordinate, which the reference's op does not do.  Values of inside points are unchanged.
    x = torch.where(inside, x, torch.zeros_like(x))
    y = torch.where(inside, y, torch.zeros_like(y))
    fx, fy = torch.floor(x), torch.floor(y)
    cx, cy = fx + 1, fy + 1
    dx, dy = cx - x, cy - y
    flat = src.reshape(B, -1)

    def tap(iy, ix):
        valid = (ix >= 0) & (ix <= Ws - 1) & (iy >= 0) & (iy <= Hs - 1)
        ixc = torch.nan_to_num(ix, nan=0.0).clamp(0, Ws - 1).long()
        iyc = torch.nan_to_num(iy, nan=0.0).clamp(0, Hs - 1).long()
        v = flat.gather(1, (iyc * Ws + ixc).reshape(B, -1)).reshape(x.shape)
        return torch.where(valid, v, torch.zeros_like(v))

    out = (dx * dy * tap(fy, fx) + (1 - dx) * (1 - dy) * tap(cy, cx)
           + dx * (1 - dy) * tap(cy, fx) + (1 - dx) * dy * tap(fy, cx))
    return torch.where(inside, out, torch.zeros_like(out))


def st_read(img: Tensor, where: Tensor, crop_size) -> Tensor:
    """Glimpse read (cell.py:135): x=(W-1)/2*(sx*X_j+tx+1), y=(H-1)/2*(sy*Y_i+ty+1).  where=[sx,tx,sy,ty]."""
    B, H, W = img.shape
    h, w = crop_size
    X = linspace_m11(w, img.dtype)[None, None, :]
    Y = linspace_m11(h, img.dtype)[None, :, None]
    sx, tx, sy, ty = (where[:, k, None, None] for k in range(4))
    x = ((sx * X + tx) + 1.0) * ((W - 1) / 2.0)
    y = ((sy * Y + ty) + 1.0) * ((H - 1) / 2.0)
    x, y = torch.broadcast_tensors(x, y)
    return resample_bilinear(img, x, y)


def st_write(glimpse: Tensor, where: Tensor, img_size) -> Tensor:
    """Inverse warp (cell.py:159): canvas pixel (I,J) samples the glimpse at
    x_g=(w-1)/2*((X_J-tx)/sx+1) computed as a'=1/sx, t'=-tx/sx (AffineGridWarper.inverse())."""
    B, h, w = glimpse.shape
    H, W = img_size
    X = linspace_m11(W, glimpse.dtype)[None, None, :]
    Y = linspace_m11(H, glimpse.dtype)[None, :, None]
    sx, tx, sy, ty = (where[:, k, None, None] for k in range(4))
    ax, ay = 1.0 / sx, 1.0 / sy
    bx, by = -tx / sx, -ty / sy
    x = ((ax * X + bx) + 1.0) * ((w - 1) / 2.0)
    y = ((ay * Y + by) + 1.0) * ((h - 1) / 2.0)
    x, y = torch.broadcast_tensors(x, y)
    return resample_bilinear(glimpse, x, y)


def st_read_gridsample(img: Tensor, where: Tensor, crop_size) -> Tensor:
    """Independent second coding of st_read via torch grid_sample(align_corners=True, zeros)."""
    B, H, W = img.shape
    h, w = crop_size
    X = linspace_m11(w, img.dtype)[None, None, :]
    Y = linspace_m11(h, img.dtype)[None, :, None]
    sx, tx, sy, ty = (where[:, k, None, None] for k in range(4))
    gx, gy = torch.broadcast_tensors(sx * X + tx, sy * Y + ty)
    grid = torch.stack([gx, gy], -1)
    return F.grid_sample(img[:, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[:, 0]


def st_write_gridsample(glimpse: Tensor, where: Tensor, img_size) -> Tensor:
    B, h, w = glimpse.shape
    H, W = img_size
    X = linspace_m11(W, glimpse.dtype)[None, None, :]
    Y = linspace_m11(H, glimpse.dtype)[None, :, None]
    sx, tx, sy, ty = (where[:, k, None, None] for k in range(4))
    gx, gy = torch.broadcast_tensors((X - tx) / sx, (Y - ty) / sy)
    grid = torch.stack([gx, gy], -1)
    return F.grid_sample(glimpse[:, None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[:, 0]


# --------------------------------------------------------------------------------------
# one AIR step (cell.py:116-171) and the unroll (model.py:66-104)
# --------------------------------------------------------------------------------------
class _SoftplusTF(torch.autograd.Function):
    """softplus with the BACKWARD in the form TF 1.1 registers for it (SoftplusGrad: gradients / (exp(-features) + 1)); the
    forward is log1p(exp(x)) (identity beyond 20, as F.softplus).  For finite gradients this equals torch's g * z / (z + 1),
    z = exp(x), to an ulp.  The forms differ only where the incoming gradient is already infinite (a KL row whose sigma^2 has
    underflowed, model.py:188-214) AND x < -88.7: TF divides inf by inf (NaN), torch multiplies inf by a denormal (inf).  Either way
    the update is non-finite; the oracle follows the reference's stack (tests/golden/ASSUMPTIONS.md #12)."""
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return F.softplus(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g / (torch.exp(-x) + 1.0)


def softplus(x: Tensor) -> Tensor:
    return _SoftplusTF.apply(x)


def transform_params(emb: Tensor, cfg: AIRConfig):
    """StochasticTransformParam._build (modules.py:58-63) + _transform (:41-46)."""
    sx, tx, sy, ty = (emb[:, k:k + 1] for k in range(4))
    loc = torch.cat([torch.sigmoid(sx), torch.tanh(tx), torch.sigmoid(sy), torch.tanh(ty)], -1)
    raw = emb[:, 4:8] + cfg.transform_var_bias
    return loc, raw


def initial_state(params, cfg: AIRConfig, obs: Tensor):
    """AIRCell.initial_state (cell.py:101-114)."""
    B = obs.shape[0]
    if cfg.transition == "lstm":
        hidden = (params["lstm/h0"].expand(B, -1), params["lstm/c0"].expand(B, -1))
    else:
        hidden = params["gru/h0"].expand(B, -1)
    z = obs.new_zeros
    return [obs.reshape(B, -1), z(B, cfg.n_pix), z(B, cfg.n_appearance), z(B, 4), hidden, obs.new_ones(B, 1)]


def _guard_scale(s: Tensor, g: float) -> Tensor:
    """scale floor of the stability switch (AIRConfig.guard_eps): max(s, g), the floored elements constant"""
    if not g or g <= 0:
        return s
    return torch.where(s < g, torch.full_like(s, g), s)


def cell_step(params, cfg: AIRConfig, state, eps_where: Tensor, eps_what: Tensor, u_pres: Tensor):
    """AIRCell._build: returns (outputs[10], new_state[6]) in the order of cell.py:167-171."""
    img_flat, canvas_flat, _what, _where, hidden, presence = state
    B = img_flat.shape[0]
    H, W = cfg.img_size
    img = img_flat.reshape(B, H, W)

    enc = mlp(img_flat, params, "input_encoder", len(cfg.inpt_encoder_hidden), last_linear=False)   # cell.py:125
    if cfg.transition == "lstm":
        h_out, c_out = lstm_step(enc, hidden[0], hidden[1], params["lstm/w_gates"], params["lstm/b_gates"])
        hidden = (h_out, c_out)
    else:
        h_out = gru_step(enc, hidden, params)
        hidden = h_out

    emb = mlp(h_out, params, "transform", len(cfg.transform_estimator_hidden) + 1, last_linear=True)
    where_loc, where_raw = transform_params(emb, cfg)                                               # cell.py:129
    where_scale = _guard_scale(softplus(where_raw), cfg.guard_eps)                                # cell.py:130-132
    where = where_loc + where_scale * eps_where                                                     # cell.py:133
    if cfg.guard_eps > 0:
        g = torch.as_tensor(cfg.guard_eps, dtype=where.dtype)
        sgn = torch.where(torch.signbit(where), -torch.ones_like(where), torch.ones_like(where))
        even = torch.tensor([True, False, True, False])
        clamped = torch.where(even & (where.abs() < g), sgn * g, where)
        where = where + (clamped - where).detach()

    cropped = st_read(img, where, cfg.crop_size)                                                    # cell.py:135

    logit = mlp(h_out, params, "steps", len(cfg.steps_pred_hidden) + 1, last_linear=True) + cfg.step_bias
    presence_prob = torch.sigmoid(logit)                                                            # cell.py:138
    if cfg.explore_eps is not None:
        presence_prob = cfg.explore_eps / 2 + (1 - cfg.explore_eps) * presence_prob                 # cell.py:140-141
    if cfg.discrete_steps:
        new_presence = (u_pres < presence_prob).to(presence_prob.dtype)                             # Bernoulli sample
        presence = presence * new_presence                                                          # cell.py:147-148
    else:
        presence = presence_prob                                                                    # cell.py:150-151

    g = mlp(cropped.reshape(B, -1), params, "glimpse_encoder", len(cfg.glimpse_encoder_hidden), last_linear=False)
    q = mm(g, params["what/w"]) + params["what/b"]                                                     # modules.py:20-21
    A = cfg.n_appearance
    what_loc, what_raw = q[:, :A], q[:, A:]
    what_scale = _guard_scale(softplus(what_raw + cfg.what_scale_offset), cfg.guard_eps)          # modules.py:23
    what = what_loc + what_scale * eps_what                                                         # cell.py:156

    decoded = mlp(what, params, "glimpse_decoder", len(cfg.glimpse_decoder_hidden) + 1, last_linear=True)
    decoded = decoded.reshape(B, *cfg.crop_size)                                                    # cell.py:158
    inversed = st_write(decoded, where, cfg.img_size)                                               # cell.py:159
    canvas_flat = canvas_flat + presence * inversed.reshape(B, -1)                                  # cell.py:164

    outputs = [canvas_flat, decoded.reshape(B, -1), what, what_loc, what_scale, where, where_loc, where_scale,
               presence_prob, presence]
    new_state = [img_flat, canvas_flat, what, where, hidden, presence]
    return outputs, new_state


OUTPUT_NAMES = "canvas glimpse what what_loc what_scale where where_loc where_scale presence_prob presence".split()


def unroll(params, cfg: AIRConfig, obs: Tensor, noise: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """AIRModel._build (model.py:66-104): time-major stacked outputs + post-processing."""
    state = initial_state(params, cfg, obs)
    per_step = []
    for t in range(cfg.max_steps):
        outs, state = cell_step(params, cfg, state, noise["eps_where"][t], noise["eps_what"][t], noise["u_pres"][t])
        per_step.append(outs)
    res = {name: torch.stack([o[i] for o in per_step], 0) for i, name in enumerate(OUTPUT_NAMES)}
    T, B = cfg.max_steps, obs.shape[0]
    hidden = state[-2]
    res["final_state"] = hidden
    res["glimpse_raw"] = res["glimpse"]
    res["glimpse"] = (res["presence"] * torch.sigmoid(res["glimpse"])).reshape(T, B, *cfg.crop_size)   # model.py:90
    res["canvas"] = res["canvas"].reshape(T, B, *cfg.img_size) * cfg.output_multiplier                  # model.py:92-93
    res["final_canvas"] = res["canvas"][-1]                                                             # model.py:95
    res["num_step_per_sample"] = res["presence"].sum(0).reshape(B)                                      # model.py:102
    res["num_step"] = res["num_step_per_sample"].mean()
    return res


# --------------------------------------------------------------------------------------
# number-of-steps prior math (prior.py) -- float64 as in the reference
# --------------------------------------------------------------------------------------
def geometric_prior(success_prob, n_steps: int, dtype=torch.float64) -> Tensor:
    """prior.py:26-32: clip to [1e-7, 1-1e-15]; Geometric(probs=1-s).prob(k) = s^k (1-s); NOT renormalised."""
    s = torch.as_tensor(success_prob, dtype=dtype).clamp(1e-7, 1.0 - 1e-15)
    k = torch.arange(n_steps + 1, dtype=dtype)
    probs = 1.0 - s
    return torch.exp(k * torch.log1p(-probs) + torch.log(probs))


def bernoulli_to_modified_geometric(presence_prob: Tensor) -> Tensor:
    """prior.py:62-68: q(n)=[1-p1, p1(1-p2), ..., prod p] in f64, renormalised, cast to f32."""
    out_dtype = torch.float32 if presence_prob.dtype != torch.float64 else torch.float64
    p = presence_prob.to(torch.float64)
    inv = 1.0 - p
    prob = torch.cumprod(p, -1)
    mod = torch.cat([inv[..., :1], inv[..., 1:] * prob[..., :-1], prob[..., -1:]], -1)
    mod = mod / mod.sum(-1, keepdim=True)
    return mod.to(out_dtype)


def tabular_kl(p: Tensor, q: Tensor, zero_prob_value: float = 0.0) -> Tensor:
    """prior.py:71-90: per-coordinate p*log(p/q) where p > zero_prob_value else 0 (masked_apply :8-23); f64 -> f32."""
    out_dtype = torch.float32 if p.dtype != torch.float64 else torch.float64
    p64, q64 = p.to(torch.float64), q.to(torch.float64)
    non_zero = p64 > zero_prob_value
    logarg = p64 / q64
    safe = torch.where(non_zero, logarg, torch.ones_like(logarg))
    log = torch.where(non_zero, torch.log(safe), torch.zeros_like(safe))
    return (p64 * log).to(out_dtype)


def clip_preserve(expr: Tensor, lo, hi) -> Tensor:
    """ops.py:67-76."""
    # tf.clip_by_value = maximum(minimum(t, clip_max), clip_min): the lower bound wins
    clipped = torch.minimum(expr, torch.as_tensor(hi, dtype=expr.dtype) if not torch.is_tensor(hi) else hi)
    clipped = torch.maximum(clipped, torch.as_tensor(lo, dtype=expr.dtype))
    return (clipped - expr).detach() + expr


def num_steps_log_prob(q: Tensor, samples: Tensor) -> Tensor:
    """NumStepsDistribution.log_prob (prior.py:143-151): gather q[b, n_b], lower clip 1e-32 (straight-through), log."""
    idx = samples.to(torch.int64).reshape(-1, 1)
    prob = q.gather(1, idx).reshape(-1)
    prob = clip_preserve(prob, 1e-32, prob.detach())
    return torch.log(prob)


def anneal_weight(init_val, final_val, anneal_type, global_step, anneal_steps, hold_for=0.0, steps_div=1.0) -> float:
    """model.py:106-124 in python floats (== float64)."""
    val, final = float(init_val), float(final_val)
    step = max(float(global_step) - float(hold_for), 0.0)
    if anneal_type == "exp":
        decay_rate = (final / val) ** (float(steps_div) / float(anneal_steps))
        val = val * decay_rate ** (step / float(steps_div))
    elif anneal_type == "linear":
        val = final + (val - final) * (1.0 - step / float(anneal_steps))
    else:
        raise NotImplementedError
    return max(final, val)


def steps_prior_success_prob(cfg: AIRConfig, global_step) -> float:
    if cfg.nsp_anneal is None:
        return float(cfg.nsp_init)
    return anneal_weight(cfg.nsp_init, cfg.nsp_final, cfg.nsp_anneal, global_step, cfg.nsp_steps,
                         cfg.nsp_hold_init, cfg.nsp_steps_div)


# --------------------------------------------------------------------------------------
# objective (model.py:126-259, 261-376)
# --------------------------------------------------------------------------------------
def normal_kl(mu_a, s_a, mu_b, s_b):
    """TF _kl_normal_normal."""
    ratio = (s_a * s_a) / (s_b * s_b)
    return (mu_a - mu_b) ** 2 / (2.0 * s_b * s_b) + 0.5 * (ratio - 1.0 - torch.log(ratio))


def baseline_forward(params, cfg: AIRConfig, obs, what, where, presence, final_state) -> Tensor:
    """BaselineMLP._build (modules.py:131-143); called with the *sampled* presence (model.py:227)."""
    B = obs.shape[0]
    parts = [t.permute(1, 0, 2).reshape(B, -1) for t in (what, where, presence)]
    parts += list(final_state) if isinstance(final_state, (tuple, list)) else [final_state]
    x = torch.cat([obs.reshape(B, -1)] + parts, -1)
    return mlp(x, params, "baseline", len(cfg.baseline_hidden) + 1, last_linear=True)           # [B,1]


def objective(params, cfg: AIRConfig, obs: Tensor, noise, global_step=0) -> Dict[str, Tensor]:
    """Forward of AIRModel.train_step's losses.  Returns every scalar / per-sample tensor the reference exposes."""
    o = unroll(params, cfg, obs, noise)
    B, T = obs.shape[0], cfg.max_steps
    dt = obs.dtype
    # reconstruction (model.py:319-324)
    mu = o["final_canvas"]
    nll = 0.5 * ((obs - mu) / cfg.output_std) ** 2 + 0.5 * math.log(2 * math.pi) + math.log(cfg.output_std)
    rec_ps = nll.sum((1, 2))
    rec = rec_ps.mean()
    res = dict(o)
    res.update(rec_loss_per_sample=rec_ps, rec_loss=rec)

    # prior (model.py:126-216)
    pp = o["presence_prob"].reshape(T, B).t()                                   # model.py:99 (squeeze + transpose)
    q = bernoulli_to_modified_geometric(pp)                                     # [B,T+1]
    s = steps_prior_success_prob(cfg, global_step)
    prior = geometric_prior(s, T)
    steps_kl = tabular_kl(q, prior[None, :])
    kl_n_ps = steps_kl.sum(1).to(dt)
    kl_n = kl_n_ps.mean()
    if cfg.nsp_analytic:
        w = torch.flip(torch.cumsum(torch.flip(q[:, 1:].t(), [0]), 0), [0]).to(dt)   # model.py:157-161  [T,B]
    else:
        w = o["presence"].reshape(T, B)
    # a prior left at None: that term is not added (model.py:174, 187)
    if cfg.what_prior is not None:
        what_kl = normal_kl(o["what_loc"], o["what_scale"], cfg.what_prior[0] * torch.ones((), dtype=dt),
                            cfg.what_prior[1] * torch.ones((), dtype=dt)).sum(-1) * w
        kl_what_ps = what_kl.sum(0)
    else:
        kl_what_ps = torch.zeros(B, dtype=dt)
    kl_what = kl_what_ps.mean()
    wl, ws = o["where_loc"], o["where_scale"]
    us, ss = wl[..., [0, 2]], ws[..., [0, 2]]                                   # model.py:190-194
    ut, st = wl[..., [1, 3]], ws[..., [1, 3]]
    one = torch.ones((), dtype=dt)
    if cfg.where_scale_prior is not None and cfg.where_shift_prior is not None:
        scale_kl = normal_kl(us, ss, cfg.where_scale_prior[0] * one, cfg.where_scale_prior[1] * one)
        # model.py:203-207: a shift prior without `loc` (here: loc = None) is centred on the posterior's own mean `ut`
        shift_mean = ut if cfg.where_shift_prior[0] is None else cfg.where_shift_prior[0] * one
        shift_kl = normal_kl(ut, st, shift_mean, cfg.where_shift_prior[1] * one)
        where_kl = (scale_kl + shift_kl).sum(-1) * w
        kl_where_ps = where_kl.sum(0)
    else:
        kl_where_ps = torch.zeros(B, dtype=dt)
    kl_where = kl_where_ps.mean()
    prior_loss = cfg.nsp_weight * kl_n + kl_what + kl_where
    prior_ps = cfg.nsp_weight * kl_n_ps + kl_what_ps + kl_where_ps
    prior_weight = 1.0 if cfg.use_prior else 0.0
    loss = rec + prior_weight * prior_loss
    loss_ps = rec_ps + prior_weight * prior_ps
    res.update(num_steps_posterior=q, steps_prior_success_prob=s, kl_num_steps_per_sample=kl_n_ps, kl_num_steps=kl_n,
               prior_step_weight=w, kl_what=kl_what, kl_where=kl_where, kl_what_per_sample=kl_what_ps,
               kl_where_per_sample=kl_where_ps, prior_loss=prior_loss, prior_loss_per_sample=prior_ps,
               loss=loss, loss_per_sample=loss_ps)

    # REINFORCE / NVIL with the reference's [B] - [B,1] -> [B,B] broadcast (model.py:218-259, SURVEY B-1)
    opt_loss = loss
    if cfg.use_reinforce:
        imp = rec_ps
        if not cfg.nsp_analytic:
            imp = imp + prior_ps
        log_prob = num_steps_log_prob(q, o["num_step_per_sample"]).to(dt)       # model.py:222
        baseline = baseline_forward(params, cfg, obs, o["what"].detach(), o["where"].detach(),
                                    o["presence"].detach(),
                                    tuple(t.detach() for t in o["final_state"]) if isinstance(o["final_state"], tuple)
                                    else o["final_state"].detach())             # [B,1]
        importance_weight = imp - baseline                                       # [B,B]: (i,j) = imp_j - b_i
        if cfg.decay_rate is not None:                                           # model.py:232-239 + ops.py:46-64
            ema = noise.setdefault("_ema", {"mean": torch.zeros((), dtype=dt), "var": torch.ones((), dtype=dt)})
            mean, var = importance_weight.detach().mean(), importance_weight.detach().var(unbiased=False)
            mm, mv = ema["mean"].clone(), ema["var"].clone()                     # variable value BEFORE this step's update
            ema["mean"] = cfg.decay_rate * ema["mean"] + (1 - cfg.decay_rate) * mean
            ema["var"] = cfg.decay_rate * ema["var"] + (1 - cfg.decay_rate) * var
            importance_weight = (importance_weight - mm) / torch.clamp(torch.sqrt(mv), min=1.0)
        reinforce_loss = (importance_weight.detach() * log_prob).mean()          # model.py:247-248
        baseline_loss = 0.5 * ((imp.detach() - baseline) ** 2).mean()            # model.py:253-256
        opt_loss = opt_loss + reinforce_loss
        if cfg.l2_weight > 0:                                                    # model.py:346-353: 2-D model weights only
            l2 = sum((v * v).sum() / 2 for k, v in params.items() if v.dim() == 2 and not is_baseline_param(k))
            res["l2_loss"] = cfg.l2_weight * l2
            opt_loss = opt_loss + res["l2_loss"]
        res.update(baseline=baseline, importance_weight=importance_weight, reinforce_loss=reinforce_loss,
                   baseline_loss=baseline_loss, num_steps_log_prob=log_prob,
                   imp_weight_mean=importance_weight.mean(), imp_weight_var=importance_weight.var(unbiased=False))
    res["opt_loss"] = opt_loss
    return res


def forward_backward(params, cfg: AIRConfig, obs, noise, global_step=0):
    """Losses + gradients exactly as the two optimisers see them (model.py:355-367): d opt_loss / d model vars and
    d baseline_loss / d baseline vars."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    res = objective(p, cfg, obs, noise, global_step)
    model_names = [k for k in p if not is_baseline_param(k)]
    base_names = [k for k in p if is_baseline_param(k)]
    gm = torch.autograd.grad(res["opt_loss"], [p[k] for k in model_names], retain_graph=True, allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(model_names, gm)}
    if cfg.use_reinforce:
        gb = torch.autograd.grad(res["baseline_loss"], [p[k] for k in base_names], allow_unused=True)
        grads.update({k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(base_names, gb)})
    else:
        grads.update({k: torch.zeros_like(p[k]) for k in base_names})
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()
            if not isinstance(v, (tuple, list))}, grads


# --------------------------------------------------------------------------------------
# optimiser: TF centred RMSProp with momentum (model.py:265)
# --------------------------------------------------------------------------------------
def rmsprop_init(params):
    return {k: dict(ms=torch.ones_like(v), mg=torch.zeros_like(v), mom=torch.zeros_like(v)) for k, v in params.items()}


def rmsprop_centered_step(params, grads, slots, cfg: AIRConfig):
    """ms<-d*ms+(1-d)g^2; mg<-d*mg+(1-d)g; mom<-m*mom+lr*g/sqrt(ms-mg^2+eps); p<-p-mom.  In place."""
    d, m, eps = cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps
    for k, p in params.items():
        lr = cfg.learning_rate * (cfg.baseline_lr_mult if is_baseline_param(k) else 1.0)
        g, s = grads[k], slots[k]
        s["ms"].mul_(d).add_(g * g, alpha=1 - d)
        s["mg"].mul_(d).add_(g, alpha=1 - d)
        denom = s["ms"] - s["mg"] * s["mg"] + eps if cfg.rms_centered else s["ms"] + eps
        s["mom"].mul_(m).add_(lr * g / torch.sqrt(denom))
        p.sub_(s["mom"])


def train_step(params, slots, cfg: AIRConfig, obs, noise, global_step=0):
    """One full reference-equivalent train step on CPU (what bench.py times as cpu_baseline)."""
    res, grads = forward_backward(params, cfg, obs, noise, global_step)
    rmsprop_centered_step(params, grads, slots, cfg)
    return res, grads
