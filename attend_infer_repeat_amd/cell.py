"""AIRCell -- one step of Attend, Infer, Repeat, with the RNNCore contract of the reference's cell
(attend_infer_repeat/cell.py:9-171): same constructor, `state_size`, `output_size`, `output_names`,
`initial_state(img)` and `cell(inpt, state) -> (outputs[10], state[6])`.  Every tensor op of the step runs in the HIP
kernels (fused GEMM layers, LSTM gates, fused ST read / canvas write, reparameterised Gaussians, presence).

Eager PyTorch replaces TF graph mode: the cell is a torch.nn.Module, `tf.nn.dynamic_rnn` becomes a python loop
(model.AIRModel) and the random draws of `.sample()` can be injected (`cell.noise`) so results can be compared with
the CPU oracle.
"""
import numpy as np
import torch

from . import functional as F
from .modules import (AffineWarpConstraints, NormalWithSoftplusScale, ParametrisedGaussian, SpatialTransformer,
                      StepsPredictor, StochasticTransformParam)


class AIRCell(torch.nn.Module):
    """RNN cell that implements the core features of Attend, Infer, Repeat (https://arxiv.org/abs/1603.08575)."""
    _n_transform_param = 4

    def __init__(self, img_size, crop_size, n_appearance,
                 transition, input_encoder, glimpse_encoder, glimpse_decoder, transform_estimator, steps_predictor,
                 discrete_steps=True, canvas_init=None, explore_eps=None, debug=False, guard_degenerate=None):
        """Arguments as in cell.py:15-36 (+ `guard_degenerate`, default None = the reference's arithmetic: a positive value floors the
        scale of both Gaussian heads and keeps the sampled scale components of `where` away from an exact zero -- the documented
        stability switch, include/air_hip.h `guard_eps`): `transition` is an RNN core instance (rnn.LSTM / rnn.GRU / anything with
        output_size, state_size, initial_state); the other five are factories called here (cell.py:61-64,69)."""
        super().__init__()
        self._img_size = tuple(int(s) for s in img_size)
        self._n_pix = int(np.prod(self._img_size))
        self._crop_size = tuple(int(s) for s in crop_size)
        self._n_appearance = int(n_appearance)
        self._transition = transition
        self._n_hidden = self._transition.output_size[0]
        self._sample_presence = discrete_steps
        self._explore_eps = explore_eps
        self._debug = debug
        self._guard_eps = float(guard_degenerate or 0.0)
        self._canvas_value = None
        if canvas_init is not None:                                          # cell.py:51-54: trainable canvas value
            self._canvas_value = torch.nn.Parameter(torch.tensor(float(canvas_init)))
        transform_constraints = AffineWarpConstraints.no_shear_2d()          # cell.py:56
        self._spatial_transformer = SpatialTransformer(img_size, crop_size, transform_constraints)
        self._inverse_transformer = SpatialTransformer(img_size, crop_size, transform_constraints, inverse=True)
        self._transform_estimator = transform_estimator(self._n_transform_param)
        self._input_encoder = input_encoder()
        self._glimpse_encoder = glimpse_encoder()
        self._glimpse_decoder = glimpse_decoder(crop_size)
        self._what_distrib = ParametrisedGaussian(n_appearance, scale_offset=0.5)   # cell.py:66
        self._what_distrib.guard_eps = self._guard_eps
        self._steps_predictor = steps_predictor()
        self.noise = None      # optional dict(eps_where[B,4], eps_what[B,A], u_pres[B,1]) consumed by the next call

    @property
    def state_size(self):                                                    # cell.py:71-80
        return [np.prod(self._img_size), np.prod(self._img_size), self._n_appearance, self._n_transform_param,
                self._transition.state_size, 1]

    @property
    def output_size(self):                                                   # cell.py:82-95
        return [np.prod(self._img_size), np.prod(self._crop_size), self._n_appearance, self._n_appearance,
                self._n_appearance, self._n_transform_param, self._n_transform_param, self._n_transform_param, 1, 1]

    @property
    def output_names(self):                                                  # cell.py:97-99
        return 'canvas glimpse what what_loc what_scale where where_loc where_scale presence_prob presence'.split()

    def _explore_eps_value(self):
        e = self._explore_eps
        if e is None:
            return None
        return float(e.item()) if torch.is_tensor(e) else float(e)

    def _validate(self, name, value, kind):
        """debug=True (cell.py:66-67,130-131,144-145: `validate_args=self._debug, allow_nan_stats=not self._debug` on the
        three distributions): TF then asserts a positive Normal scale and Bernoulli probabilities inside [0, 1], and NaNs
        trip those assertions.  Eager counterpart: a device sync per check, so only in debug mode."""
        if not self._debug:
            return
        ok = torch.isfinite(value).all()
        if kind == "scale":
            ok = ok & (value > 0).all()
        elif kind == "prob":
            ok = ok & (value >= 0).all() & (value <= 1).all()
        if not bool(ok):
            raise ValueError("AIRCell(debug=True): invalid %s parameter `%s` (non-finite or outside its support)"
                             % ("Normal" if kind != "prob" else "Bernoulli", name))

    def initial_state(self, img):                                            # cell.py:101-114
        batch_size = img.shape[0]
        dev = img.device
        hidden_state = self._transition.initial_state(batch_size, torch.float32, trainable=True, device=dev)
        where_code = torch.zeros(batch_size, self._n_transform_param, device=dev)
        what_code = torch.zeros(batch_size, self._n_appearance, device=dev)
        flat_canvas = torch.zeros(batch_size, self._n_pix, device=dev)
        if self._canvas_value is not None:
            flat_canvas = flat_canvas + self._canvas_value
        flat_img = img.reshape(batch_size, self._n_pix)
        init_presence = torch.ones(batch_size, 1, device=dev)
        return [flat_img, flat_canvas, what_code, where_code, hidden_state, init_presence]

    def forward(self, inpt, state):
        """Input is unused; it only forces a maximum number of steps (cell.py:116-117)."""
        img_flat, canvas_flat, what_code, where_code, hidden_state, presence = state
        B = img_flat.shape[0]
        img = img_flat.reshape((B,) + self._img_size)
        noise = self.noise or {}
        self.noise = None

        inpt_encoding = self._input_encoder(img)                             # cell.py:125
        hidden_output, hidden_state = self._transition(inpt_encoding, hidden_state)

        est = self._transform_estimator                                      # cell.py:129-133
        if isinstance(est, StochasticTransformParam):
            loc_pre, raw = est.embed_split(hidden_output)
            where_distrib = NormalWithSoftplusScale(loc_pre, raw, raw_offset=est.scale_bias, loc_mode=1,
                                                    guard_eps=self._guard_eps)
        else:
            where_distrib = NormalWithSoftplusScale(*est(hidden_output))
        where_code = where_distrib.sample(noise.get("eps_where"))
        where_loc, where_scale = where_distrib.loc, where_distrib.scale
        self._validate("where_loc", where_loc, "loc"); self._validate("where_scale", where_scale, "scale")

        cropped = self._spatial_transformer(img, where_code)                 # cell.py:135

        pred = self._steps_predictor                                         # cell.py:137-151
        eps = self._explore_eps_value()
        if isinstance(pred, StepsPredictor):
            u = noise.get("u_pres")
            if u is None and self._sample_presence:
                u = torch.rand(B, 1, device=img.device)
            logit = pred.logit(hidden_output).reshape(1, B)
            prob, pres = F.presence(logit, None if u is None else u.reshape(1, B), presence.reshape(B),
                                    pred.steps_bias, eps, self._sample_presence)
            presence_prob, presence = prob.reshape(B, 1), pres.reshape(B, 1)
        else:
            presence_prob = pred(hidden_output)
            if eps is not None:
                presence_prob = eps / 2 + (1 - eps) * presence_prob
            if self._sample_presence:
                u = noise.get("u_pres")
                u = torch.rand_like(presence_prob) if u is None else u
                presence = presence * (u < presence_prob).to(presence_prob.dtype)
            else:
                presence = presence_prob

        self._validate("presence_prob", presence_prob, "prob")
        what_params = self._glimpse_encoder(cropped)                         # cell.py:153-156
        what_distrib = self._what_distrib(what_params)
        what_code = what_distrib.sample(noise.get("eps_what"))
        what_loc, what_scale = what_distrib.loc, what_distrib.scale
        self._validate("what_loc", what_loc, "loc"); self._validate("what_scale", what_scale, "scale")

        decoded = self._glimpse_decoder(what_code)                           # cell.py:158-165 (one fused kernel)
        canvas = F.st_write_acc(decoded.reshape((B,) + self._crop_size), where_code, presence,
                                canvas_flat.reshape((B,) + self._img_size), self._img_size)
        canvas_flat = canvas.reshape(B, self._n_pix)
        decoded_flat = decoded.reshape(B, int(np.prod(self._crop_size)))

        output = [canvas_flat, decoded_flat, what_code, what_loc, what_scale, where_code, where_loc, where_scale,
                  presence_prob, presence]
        state = [img_flat, canvas_flat, what_code, where_code, hidden_state, presence]
        return output, state
