"""AIRonMNIST -- AIR for the multi-MNIST dataset with the constructor of the reference
(attend_infer_repeat/mnist_model.py:10-44): n_appearance=50, LSTM(256) transition, output_std=.3, stock modules.

Because this is the standard architecture, `train_step` runs through the fused hipGraph-captured engine
(engine.AIREngine) instead of autograd; the module tree shares its parameters with the engine's flat buffer, so
cell-by-cell calls and `forward()` always see the trained weights.
"""
from functools import partial

import torch

from .engine import AIREngine, EngineConfig
from .model import AIRModel
from .modules import BaselineMLP, Decoder, Encoder, StepsPredictor, StochasticTransformParam
from .rnn import LSTM


class AIRonMNIST(AIRModel):
    """Implements AIR for the MNIST dataset"""

    def __init__(self, obs, nums, glimpse_size=(20, 20),
                 inpt_encoder_hidden=[256] * 2,
                 glimpse_encoder_hidden=[256] * 2,
                 glimpse_decoder_hidden=[252] * 2,
                 transform_estimator_hidden=[256] * 2,
                 steps_pred_hidden=[50] * 1,
                 baseline_hidden=[256, 128] * 1,
                 transform_var_bias=-2.,
                 step_bias=0.,
                 *args, **kwargs):
        self.transform_var_bias = torch.tensor(float(transform_var_bias))     # non-trainable variables,
        self.step_bias = torch.tensor(float(step_bias))                       # mnist_model.py:24-26
        self.baseline = BaselineMLP(baseline_hidden)
        self._hyper = dict(inpt_encoder_hidden=tuple(inpt_encoder_hidden),
                           glimpse_encoder_hidden=tuple(glimpse_encoder_hidden),
                           glimpse_decoder_hidden=tuple(glimpse_decoder_hidden),
                           transform_estimator_hidden=tuple(transform_estimator_hidden),
                           steps_pred_hidden=tuple(steps_pred_hidden), baseline_hidden=tuple(baseline_hidden))
        super(AIRonMNIST, self).__init__(
            *args,
            obs=obs,
            nums=nums,
            glimpse_size=glimpse_size,
            n_appearance=50,
            transition=LSTM(256),
            input_encoder=partial(Encoder, inpt_encoder_hidden),
            glimpse_encoder=partial(Encoder, glimpse_encoder_hidden),
            glimpse_decoder=partial(Decoder, glimpse_decoder_hidden),
            transform_estimator=partial(StochasticTransformParam, transform_estimator_hidden,
                                        scale_bias=self.transform_var_bias),
            steps_predictor=partial(StepsPredictor, steps_pred_hidden, self.step_bias),
            output_std=.3,
            **kwargs
        )

    # ---- name map between the module tree and the engine's flat parameter buffer -------------------------------------
    def _named_module_params(self):
        c = self.cell
        out = {}

        def mlp(prefix, m):
            for i, layer in enumerate(m.layers):
                out[f"{prefix}/{i}/w"], out[f"{prefix}/{i}/b"] = layer.w, layer.b

        mlp("input_encoder", c._input_encoder.mlp)
        out["lstm/w_gates"], out["lstm/b_gates"] = c._transition.w_gates, c._transition.b_gates
        out["lstm/h0"], out["lstm/c0"] = c._transition.h0, c._transition.c0
        mlp("transform", c._transform_estimator.mlp)
        mlp("steps", c._steps_predictor.mlp)
        mlp("glimpse_encoder", c._glimpse_encoder.mlp)
        out["what/w"], out["what/b"] = c._what_distrib.w, c._what_distrib.b
        mlp("glimpse_decoder", c._glimpse_decoder.mlp)
        bm = getattr(self, "baseline_module", None)
        if bm is not None and all(l.w is not None for l in bm.mlp.layers):   # built lazily by the first _reinforce call
            mlp("baseline", bm.mlp)
        return out

    def _engine_eligible(self, use_engine, l2_weight, what_prior, where_scale_prior, where_shift_prior, num_steps_prior,
                         decay_rate):
        """The fused engine takes the reference script's configuration (scripts/multi_mnist.py:24-94) and, since round 5, the rest
        of train_step's plain arguments (model.py:261-353): l2_weight, decay_rate (EMA-normalised importance weights), a weighted
        num-steps prior, a where-shift prior without `loc`, the RMSProp keyword set (decay / momentum / epsilon / centered), a
        what / where prior left at None (model.py:174, 187: the term is not added) and a non-analytic num-steps prior (sampled step
        weights, the prior inside the importance weight: model.py:157-163, 339-340).
        Continuous steps (discrete_steps=False, cell.py:150-151: the presence is the probability itself and carries a gradient
        through the canvas write) run on the engine too.  What still trains through the generic autograd path over the same kernels:
        a custom optimizer CLASS and a non-MLP baseline.  num_steps_prior=None is an error in the reference too (model.py:157 reads
        its `analytic`)."""
        nsp = num_steps_prior
        has = lambda p, *keys: p is not None and all(k in p for k in keys)
        if not use_engine:
            return False
        if getattr(self, "_custom_optimizer", None) is not None:
            return False
        if nsp is None:
            return False
        if what_prior is not None and not has(what_prior, 'loc', 'scale'):
            return False
        if where_scale_prior is not None and where_shift_prior is not None and not (
                has(where_scale_prior, 'loc', 'scale') and has(where_shift_prior, 'scale')):
            return False
        if decay_rate is not None and not self.use_reinforce:
            return False
        if self.use_reinforce:
            bm = getattr(self, "baseline_module", None)
            if not isinstance(bm, BaselineMLP) or any(l.w is None for l in bm.mlp.layers):
                return False
        return True

    def engine_config(self, learning_rate, num_steps_prior, what_prior, where_scale_prior, where_shift_prior, l2_weight=0.,
                      decay_rate=None):
        nsp = num_steps_prior
        rms = getattr(self, "_rms_kwargs", None) or dict(decay=0.9, momentum=0.9, epsilon=1e-10, centered=True)
        has_where = where_scale_prior is not None and where_shift_prior is not None
        return EngineConfig(
            img_size=tuple(self.img_size), crop_size=tuple(self.glimpse_size), n_appearance=self.n_appearance,
            n_hidden=256, max_steps=self.max_steps,
            transform_var_bias=float(self.transform_var_bias), step_bias=float(self.step_bias),
            output_multiplier=float(self.output_multiplier), output_std=float(self.output_std),
            explore_eps=None if self.explore_eps is None else float(self.explore_eps),
            what_prior=None if what_prior is None else (what_prior.loc, what_prior.scale),
            where_scale_prior=None if not has_where else (where_scale_prior.loc, where_scale_prior.scale),
            where_shift_prior=None if not has_where else (where_shift_prior.loc if 'loc' in where_shift_prior else None,
                                                          where_shift_prior.scale),
            nsp_analytic=bool(getattr(nsp, 'analytic', True)), discrete_steps=bool(self.discrete_steps),
            nsp_anneal=getattr(nsp, 'anneal', None), nsp_init=nsp.init, nsp_final=getattr(nsp, 'final', nsp.init),
            nsp_steps_div=getattr(nsp, 'steps_div', 1.), nsp_steps=getattr(nsp, 'steps', 1.),
            nsp_hold_init=getattr(nsp, 'hold_init', 0.),
            use_prior=self.use_prior, use_reinforce=self.use_reinforce, learning_rate=float(learning_rate),
            guard_eps=float(getattr(self, 'guard_degenerate', 0.0)),
            l2_weight=float(l2_weight or 0.), decay_rate=None if decay_rate is None else float(decay_rate),
            nsp_weight=float(getattr(nsp, 'weight', 1.)), rms_decay=float(rms["decay"]), rms_momentum=float(rms["momentum"]),
            rms_eps=float(rms["epsilon"]), rms_centered=bool(rms["centered"]), **self._hyper)

    def train_step(self, learning_rate, l2_weight=0., what_prior=None, where_scale_prior=None,
                   where_shift_prior=None, num_steps_prior=None, use_prior=True, use_reinforce=True, baseline=None,
                   decay_rate=None, optimizer=None, opt_kwargs=None, use_engine=True, capture_graph=True,
                   mfma_dtype="f32"):
        """model.py:261-376 on the fused engine.  Extras over the reference signature: `use_engine` / `capture_graph`, and
        `mfma_dtype` ("f32" exact fp32 MFMA, "bf16" = bf16-rounded operands with fp32 accumulate in every dense product)."""
        fn, gs = super(AIRonMNIST, self).train_step(learning_rate, l2_weight, what_prior, where_scale_prior,
                                                    where_shift_prior, num_steps_prior, use_prior, use_reinforce,
                                                    baseline, decay_rate, optimizer, opt_kwargs)
        if not self._engine_eligible(use_engine, l2_weight, what_prior, where_scale_prior, where_shift_prior,
                                     num_steps_prior, decay_rate):
            return fn, gs
        self._hyper["mfma_dtype"] = mfma_dtype
        cfg = self.engine_config(learning_rate, num_steps_prior, what_prior, where_scale_prior, where_shift_prior, l2_weight,
                                 decay_rate)
        eng = AIREngine(cfg, self.batch_size, device=self.obs.device)
        named = self._named_module_params()
        eng.load_parameters({k: v.detach() for k, v in named.items()})
        eng.synchronize()
        for k, p in named.items():                           # share storage: modules now view the engine's flat buffer
            p.data = eng.params[k]
        eng.set_obs(self.obs)
        if capture_graph:
            eng.capture()
        self._engine = eng

        state = {"lr": None}

        def train_step_fn(obs=None, nums=None, refresh=True):
            """One fused update (fresh noise, forward, backward, both RMSProp updates) as a hipGraph replay.
            refresh=False skips re-exposing the engine buffers as model attributes (a device sync + a handful of torch
            ops per call): use it in tight training loops and call `air.refresh()` / `air.evaluate(...)` when needed."""
            if obs is not None:
                self.obs = obs
            if nums is not None:
                self.nums = nums
            lr = float(self.learning_rate)
            if lr != state["lr"]:
                eng.set_learning_rate(lr); state["lr"] = lr
            self._sync_engine_switches()
            eng.train_step(obs)
            self.global_step += 1
            if refresh:
                self._refresh_from_engine()
            return self.global_step

        self._train_step = train_step_fn
        return self._train_step, self.global_step

    def _sync_engine_switches(self):
        """The reference's non-trainable variables (use_prior / toggle_prior, explore_eps, step_bias, transform_var_bias,
        output_multiplier: model.py:58,71,307-308, mnist_model.py:24-26) are plain attributes here; whatever they hold NOW is
        what the next engine launch uses (AIREngine.update_config re-captures when one of them changed)."""
        eng = self._engine
        if eng is not None:
            eng.update_config(use_prior=bool(self.use_prior),
                              explore_eps=None if self.explore_eps is None else float(self.explore_eps),
                              step_bias=float(self.step_bias), transform_var_bias=float(self.transform_var_bias),
                              output_multiplier=float(self.output_multiplier))

    def forward(self, obs=None, nums=None, noise=None):
        """Generic cell-by-cell unroll (model.py:66-104).  Once the engine owns the parameters the module tree aliases its
        flat buffer, so this torch-stream pass is ordered after the engine's pending updates, and the engine's next launch
        after this pass."""
        eng = getattr(self, "_engine", None)
        if eng is not None:
            eng.wait_for_engine()
        out = super(AIRonMNIST, self).forward(obs, nums, noise)
        if eng is not None:
            eng.wait_for_caller()
        return out

    def evaluate(self, obs=None, nums=None, noise=None):
        """Engine-backed evaluation pass (fresh noise, no update); falls back to the generic path without an engine."""
        if self._engine is None:
            return super(AIRonMNIST, self).evaluate(obs, nums, noise)
        if obs is not None:
            self.obs = obs
        if nums is not None:
            self.nums = nums
        self._sync_engine_switches()
        self._engine.forward(self.obs, sample_noise=True)
        self._refresh_from_engine()
        return self

    def refresh(self):
        """Re-expose the engine's current buffers under the reference's attribute names."""
        if self._engine is not None:
            self._refresh_from_engine()
        return self

    def _refresh_from_engine(self):
        """Expose the engine's buffers under the reference's attribute names (model.py:86-104,319-343)."""
        eng, T, B = self._engine, self.max_steps, self.batch_size
        o = eng.outputs()
        for k in ("what", "what_loc", "what_scale", "where", "where_loc", "where_scale", "presence_prob", "presence",
                  "glimpse", "canvas", "final_canvas", "final_state", "rec_loss_per_sample", "rec_loss",
                  "kl_num_steps_per_sample", "kl_num_steps", "kl_what", "kl_where", "prior_step_weight",
                  "num_step_per_sample", "opt_loss", "reinforce_loss", "baseline_loss", "baseline"):
            if k == "kl_what" and eng.cfg.what_prior is None or k == "kl_where" and eng.cfg.where_scale_prior is None:
                continue                                   # model.py:174, 187: a prior left at None defines no such tensor
            if k in o:
                setattr(self, k, o[k])
        self.num_step = self.num_step_per_sample.mean()
        from .ops import Loss
        from .prior import NumStepsDistribution
        self.prior_loss = Loss(); self.prior_loss.add(o["prior_loss"], float(eng.cfg.nsp_weight) * o["kl_num_steps_per_sample"]
                                                      + o["kl_what_per_sample"] + o["kl_where_per_sample"])
        for k in ("l2_loss", "imp_weight_moving_mean", "imp_weight_moving_var"):
            if k in o:
                setattr(self, k, o[k])
        self.loss = Loss(); self.loss.add(o["loss"], o["rec_loss_per_sample"]
                                          + self.prior_weight * self.prior_loss.per_sample)
        if "baseline" in o:
            imp = o["rec_loss_per_sample"] if eng.cfg.nsp_analytic else o["rec_loss_per_sample"] + self.prior_loss.per_sample
            self.reinforce_imp_weight = imp                                                   # model.py:337-340
            self.importance_weight = imp[None, :] - o["baseline"]                             # [B,B] quirk, model.py:230
        self.num_steps_distrib = NumStepsDistribution(o["presence_prob"].reshape(T, B).t())
        self.steps_prior_success_prob = eng.steps_prior_success_prob(max(eng.global_step - 1, 0))
        if self.nums is not None:
            self.gt_num_steps = self.nums.sum(0).reshape(-1)
            self.num_step_accuracy = (self.gt_num_steps == self.num_step_per_sample).float().mean()
