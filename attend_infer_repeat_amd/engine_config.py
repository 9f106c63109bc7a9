"""Hyper-parameters, the flat parameter layout and the host-side schedule helpers of the fused engine (engine.py)."""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


@dataclass
class EngineConfig:
    """Hyper-parameters of the standard AIR-on-MNIST architecture (mnist_model.py:13-44, scripts/multi_mnist.py:24-94)."""
    img_size: Tuple[int, int] = (50, 50)
    crop_size: Tuple[int, int] = (20, 20)
    n_appearance: int = 50
    n_hidden: int = 256
    inpt_encoder_hidden: Sequence[int] = (256, 256)
    glimpse_encoder_hidden: Sequence[int] = (256, 256)
    glimpse_decoder_hidden: Sequence[int] = (256, 256)
    transform_estimator_hidden: Sequence[int] = (256, 256)
    steps_pred_hidden: Sequence[int] = (128, 64)
    baseline_hidden: Sequence[int] = (256, 128)
    max_steps: int = 3
    transform_var_bias: float = 0.5
    step_bias: float = 0.75
    output_multiplier: float = 0.5
    output_std: float = 0.3
    explore_eps: Optional[float] = 1e-3
    what_scale_offset: float = 0.5
    what_prior: Optional[Tuple[float, float]] = (0.0, 1.0)              # None: the term is not added (model.py:174)
    where_scale_prior: Optional[Tuple[float, float]] = (0.0, 1.0)       # either of the two at None: no where term (model.py:187)
    where_shift_prior: Optional[Tuple[Optional[float], float]] = (0.0, 1.0)   # loc None: centred on the posterior's own mean
    nsp_anneal: Optional[str] = "exp"
    nsp_init: float = 1.0 - 1e-15
    nsp_final: float = 1e-7
    nsp_steps_div: float = 1e4
    nsp_steps: float = 1e5
    nsp_hold_init: float = 1e3
    use_prior: bool = True
    use_reinforce: bool = True
    learning_rate: float = 1e-4
    baseline_lr_mult: float = 10.0
    rms_decay: float = 0.9
    rms_momentum: float = 0.9
    rms_eps: float = 1e-10
    rms_centered: bool = True             # opt_kwargs of model.py:265 (centered=False: tf.train.RMSPropOptimizer's own default)
    # the rest of train_step's arguments (model.py:261-353), all off in the script:
    l2_weight: float = 0.0                # l2_weight * sum(w^2) / 2 over the 2-D model variables (model.py:346-353)
    decay_rate: Optional[float] = None    # EMA normalisation of the importance weight (model.py:232-239, ops.py:46-64)
    nsp_weight: float = 1.0               # num_steps_prior.weight (model.py:339-340)
    discrete_steps: bool = True         # False (cell.py:150-151): presence = presence_prob, with a gradient through the canvas write
    nsp_analytic: bool = True             # False: step weights = the sampled presences, the prior joins the importance weight (model.py:157-163,339-340)
    # what_prior / where_scale_prior + where_shift_prior may be None: that KL term is then left out of the loss (model.py:174-209)
    # "f32": exact fp32 MFMA everywhere.  "bf16": every dense product (MLPs, LSTM gates, their dX / dW) rounds its operands
    # to bf16 in registers and multiplies on the bf16 MFMA with fp32 accumulate; parameters, activations, gradients and the
    # optimiser stay fp32 (BASELINE.json configs[4], "bf16 MFMA MLP path").
    mfma_dtype: str = "f32"
    # Stability switch, default off (= the reference's arithmetic, inf / NaN placement included): > 0 floors the scale of both
    # Gaussian heads at this value and keeps the sampled scale components of `where` at |s| >= it (include/air_hip.h `guard_eps`;
    # model.py:188-214 has no clamp, cell.py:130-133 can sample an exact zero scale)
    guard_eps: float = 0.0

    @property
    def n_pix(self):
        return int(self.img_size[0] * self.img_size[1])

    @property
    def n_crop(self):
        return int(self.crop_size[0] * self.crop_size[1])

    @property
    def baseline_in(self):
        T = self.max_steps
        return self.n_pix + T * self.n_appearance + T * 4 + T + 2 * self.n_hidden


def _mlp_shapes(n_in, hiddens, n_out):
    sizes = list(hiddens) + ([n_out] if n_out is not None else [])
    out, prev = [], n_in
    for s in sizes:
        out.append((prev, int(s)))
        prev = int(s)
    return out


def param_shapes(cfg: EngineConfig) -> "Dict[str, Tuple[int, ...]]":
    """Ordered name -> shape of every trainable tensor; model variables first, baseline variables last (the two
    optimisers of model.py:355-367 each own one contiguous segment of the flat buffer)."""
    Hd, A = cfg.n_hidden, cfg.n_appearance
    out: Dict[str, Tuple[int, ...]] = {}

    def add(prefix, n_in, hiddens, n_out):
        for i, (a, b) in enumerate(_mlp_shapes(n_in, hiddens, n_out)):
            out[f"{prefix}/{i}/w"] = (a, b)
            out[f"{prefix}/{i}/b"] = (b,)

    add("input_encoder", cfg.n_pix, cfg.inpt_encoder_hidden, None)
    enc_out = int(cfg.inpt_encoder_hidden[-1])
    out["lstm/w_gates"] = (enc_out + Hd, 4 * Hd)
    out["lstm/b_gates"] = (4 * Hd,)
    out["lstm/h0"] = (1, Hd)
    out["lstm/c0"] = (1, Hd)
    add("transform", Hd, cfg.transform_estimator_hidden, 8)
    add("steps", Hd, cfg.steps_pred_hidden, 1)
    add("glimpse_encoder", cfg.n_crop, cfg.glimpse_encoder_hidden, None)
    out["what/w"] = (int(cfg.glimpse_encoder_hidden[-1]), 2 * A)
    out["what/b"] = (2 * A,)
    add("glimpse_decoder", A, cfg.glimpse_decoder_hidden, cfg.n_crop)
    add("baseline", cfg.baseline_in, cfg.baseline_hidden, 1)
    return out


def anneal_weight(init_val, final_val, anneal_type, global_step, anneal_steps, hold_for=0.0, steps_div=1.0):
    """model.py:106-124 (float64 == python float)."""
    val, final = float(init_val), float(final_val)
    step = max(float(global_step) - float(hold_for), 0.0)
    if anneal_type == "exp":
        decay_rate = (final / val) ** (float(steps_div) / float(anneal_steps))
        val = val * decay_rate ** (step / float(steps_div))
    elif anneal_type == "linear":
        val = final + (val - final) * (1.0 - step / float(anneal_steps))
    else:
        raise NotImplementedError(anneal_type)
    return max(final, val)


def geometric_prior_f64(success_prob: float, n_steps: int) -> List[float]:
    """prior.py:26-32: clip, Geometric(probs=1-s).prob(k) = exp(k*log1p(-probs) + log(probs)); not renormalised."""
    s = min(max(float(success_prob), 1e-7), 1.0 - 1e-15)
    probs = 1.0 - s
    return [math.exp(k * math.log1p(-probs) + math.log(probs)) for k in range(n_steps + 1)]


class _Mlp:
    """bookkeeping for one MLP: weights/bias/grad views, activation buffers"""

    def __init__(self, eng, prefix, n_rows, n_in, hiddens, n_out):
        self.prefix = prefix
        self.shapes = _mlp_shapes(n_in, hiddens, n_out)
        self.n = len(self.shapes)
        self.last_linear = n_out is not None
        self.w = [eng.params[f"{prefix}/{i}/w"] for i in range(self.n)]
        self.b = [eng.params[f"{prefix}/{i}/b"] for i in range(self.n)]
        self.dw = [eng.grads[f"{prefix}/{i}/w"] for i in range(self.n)]
        self.db = [eng.grads[f"{prefix}/{i}/b"] for i in range(self.n)]
        self.out = [eng._buf(f"{prefix}/act{i}", (n_rows, s[1])) for i, s in enumerate(self.shapes)]
        self.g = [eng._buf(f"{prefix}/g{i}", (n_rows, s[1])) for i, s in enumerate(self.shapes)]
        self.rows = n_rows
