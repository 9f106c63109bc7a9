"""AIRModel -- the generic AIR model with the attribute surface of the reference's model
(attend_infer_repeat/model.py:15-376), eager PyTorch on HIP kernels.

Differences forced by eager mode: tensors are concrete; `AIRModel(obs, ...)` runs the unroll on
`obs` right away; `train_step(...)` returns `(train_step_fn, global_step)` where `train_step_fn(obs=None, nums=None)`
performs ONE update (what `sess.run(train_step)` did) and refreshes every attribute; `global_step` is a 0-dim int64
tensor.  For the standard architecture (LSTM transition + the stock modules) the update runs through the fused,
hipGraph-captured engine (engine.AIREngine); any other architecture trains through autograd over the same kernels.
"""
import functools
import math

import numpy as np
import torch

from . import functional as F
from . import hip as H
from .cell import AIRCell
from .ops import Loss, make_moving_average
from .prior import NumStepsDistribution, geometric_prior, tabular_kl


class _OutputDistrib(object):
    """Normal(final_canvas, output_std) (model.py:97): only what the model itself uses."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, float(scale)

    def log_prob(self, x):
        return -(0.5 * ((x - self.loc) / self.scale) ** 2 + 0.5 * math.log(2 * math.pi) + math.log(self.scale))


class AIRModel(object):
    """Generic AIR model"""

    @property
    def explore_eps(self):
        """model.py:70-71: a non-trainable variable SHARED with the cell (cell.py:47,140-141) -- assigning it here changes
        what the next unroll clips the step probability with, on the generic path and (through
        AIRonMNIST._sync_engine_switches) on the engine."""
        return self._explore_eps

    @explore_eps.setter
    def explore_eps(self, value):
        self._explore_eps = value
        cell = getattr(self, "cell", None)
        if cell is not None:
            cell._explore_eps = value

    def __init__(self, obs, nums, max_steps, glimpse_size,
                 n_appearance, transition, input_encoder, glimpse_encoder, glimpse_decoder, transform_estimator,
                 steps_predictor,
                 output_std=1., discrete_steps=True, output_multiplier=1.,
                 explore_eps=None, debug=False, **kwargs):
        """Arguments as in model.py:18-43."""
        self.obs = obs
        self.nums = nums
        self.max_steps = int(max_steps)
        self.glimpse_size = tuple(glimpse_size)
        self.n_appearance = n_appearance
        self.output_std = output_std
        self.discrete_steps = discrete_steps
        self.explore_eps = explore_eps
        self.debug = debug
        self.guard_degenerate = float(kwargs.get('guard_degenerate') or 0.0)     # (extension, default off: AIRCell docstring)
        self.output_multiplier = torch.tensor(float(output_multiplier))        # non-trainable variable, model.py:58
        shape = list(self.obs.shape)
        self.batch_size = shape[0]
        self.img_size = shape[1:]
        self._engine = None
        self._moving_averages = {}           # state of make_moving_average (decay_rate), owned by this model
        self._in_train_step = False
        self._build(transition, input_encoder, glimpse_encoder, glimpse_decoder, transform_estimator,
                    steps_predictor, kwargs)

    # ---- model.py:66-104 -------------------------------------------------------------------------------------------
    def _build(self, transition, input_encoder, glimpse_encoder, glimpse_decoder, transform_estimator,
               steps_predictor, kwargs):
        self.cell = AIRCell(self.img_size, self.glimpse_size, self.n_appearance, transition,
                            input_encoder, glimpse_encoder, glimpse_decoder, transform_estimator, steps_predictor,
                            canvas_init=None, discrete_steps=self.discrete_steps, explore_eps=self.explore_eps,
                            debug=self.debug, **kwargs)
        self.forward()

    def forward(self, obs=None, nums=None, noise=None):
        """(Re-)runs the T-step unroll (tf.nn.dynamic_rnn at model.py:83-84) and refreshes the output attributes.
        noise: optional dict of time-major eps_where[T,B,4], eps_what[T,B,A], u_pres[T,B,1]."""
        if obs is not None:
            self.obs = obs
        if nums is not None:
            self.nums = nums
        T, B = self.max_steps, self.batch_size
        state = self.cell.initial_state(self.obs)
        outs = []
        for t in range(T):
            if noise is not None:
                self.cell.noise = {k: v[t] for k, v in noise.items()}
            o, state = self.cell(None, state)
            outs.append(o)
        for i, name in enumerate(self.cell.output_names):
            setattr(self, name, torch.stack([o[i] for o in outs], 0))
        self.final_state = state[-2]
        mult = float(self.output_multiplier)
        self.glimpse = (self.presence * torch.sigmoid(self.glimpse)).reshape((T, B) + tuple(self.glimpse_size))
        self._canvas_unscaled = self.canvas
        self.canvas = self.canvas.reshape((T, B) + tuple(self.img_size)) * mult
        self.final_canvas = self.canvas[-1]
        self.output_distrib = _OutputDistrib(self.final_canvas, self.output_std)
        posterior_step_probs = self.presence_prob.reshape(T, B).t()
        self.num_steps_distrib = NumStepsDistribution(posterior_step_probs)
        self.num_step_per_sample = self.presence.sum(0).reshape(B).float()
        self.num_step = self.num_step_per_sample.mean()
        if self.nums is not None:
            self.gt_num_steps = self.nums.sum(0).reshape(-1)
        return self

    # ---- model.py:106-124 ------------------------------------------------------------------------------------------
    @staticmethod
    def _anneal_weight(init_val, final_val, anneal_type, global_step, anneal_steps, hold_for=0., steps_div=1.,
                       dtype=torch.float64):
        val, final, step, hold_for, anneal_steps, steps_div = (float(i) for i in (init_val, final_val, global_step,
                                                                                   hold_for, anneal_steps, steps_div))
        step = max(step - hold_for, 0.)
        if anneal_type == 'exp':
            decay_rate = (final / val) ** (steps_div / anneal_steps)
            val = val * decay_rate ** (step / steps_div)
        elif anneal_type == 'linear':
            val = final + (val - final) * (1. - step / anneal_steps)
        else:
            raise NotImplementedError
        return max(final, val)

    # ---- model.py:126-216 ------------------------------------------------------------------------------------------
    def _prior_loss(self, what_prior, where_scale_prior, where_shift_prior, num_steps_prior, global_step):
        T, B = self.max_steps, self.batch_size
        prior_loss = Loss()
        nsp = num_steps_prior
        if nsp is not None:
            if getattr(nsp, 'anneal', None) is not None:
                s = self._anneal_weight(nsp.init, nsp.final, nsp.anneal, int(global_step), nsp.steps,
                                        getattr(nsp, 'hold_init', 0.), getattr(nsp, 'steps_div', 1.))
            else:
                s = nsp.init
            self.steps_prior_success_prob = s
            prior = geometric_prior(s, self.max_steps).to(self.obs.device)
            pp = self.presence_prob.reshape(T, B)
            q, kl_ps, logp, w = F.numsteps(pp, self.presence.reshape(T, B), prior.double())
            self._num_steps_q, self._num_steps_logp = q, logp
            self.kl_num_steps_per_sample = kl_ps
            self.kl_num_steps = kl_ps.mean()
            prior_loss.add(self.kl_num_steps, self.kl_num_steps_per_sample, weight=getattr(nsp, 'weight', 1.))
        if getattr(nsp, 'analytic', True):
            step_weight = w                                                   # sum_{n>t} q(n), model.py:157-161
        else:
            step_weight = self.presence.reshape(T, B)
        self.prior_step_weight = step_weight
        if what_prior is not None:
            what_kl = F.normal_kl_rows(self.what_loc, self.what_scale,
                                       (what_prior.loc, what_prior.scale, what_prior.loc, what_prior.scale))
            what_kl_per_sample = (what_kl * step_weight).sum(0)
            self.kl_what = what_kl_per_sample.mean()
            prior_loss.add(self.kl_what, what_kl_per_sample, weight=1.)
        if where_scale_prior is not None and where_shift_prior is not None:
            own_mean = 'loc' not in where_shift_prior
            shift_mean = 0. if own_mean else where_shift_prior.loc
            where_kl = F.normal_kl_rows(self.where_loc, self.where_scale,
                                        (where_scale_prior.loc, where_scale_prior.scale, shift_mean,
                                         where_shift_prior.scale))
            if own_mean:
                # model.py:203-207: without `loc` the shift prior is centred on the posterior's own mean `ut`, so the
                # (mu_a - mu_b)^2 / (2 s_b^2) term of the KL -- and its gradient, d/d ut of (ut - ut)^2 -- vanishes:
                # remove exactly that term from the rows evaluated against a zero mean
                ut = self.where_loc[..., 1::2]
                where_kl = where_kl - (ut * ut).sum(-1) / (2. * float(where_shift_prior.scale) ** 2)
            where_kl_per_sample = (where_kl * step_weight).sum(0)
            self.kl_where = where_kl_per_sample.mean()
            prior_loss.add(self.kl_where, where_kl_per_sample, weight=1.)
        return prior_loss

    # ---- model.py:218-259 ------------------------------------------------------------------------------------------
    def _reinforce(self, importance_weight, decay_rate):
        log_prob = self._num_steps_logp                       # log q(n = sum_t presence_t), model.py:222
        if self.baseline is None:
            self.baseline = torch.zeros(self.batch_size, 1, device=self.obs.device)       # no learned baseline
            self.baseline_module = None
        elif getattr(self, 'baseline_module', None) is not None or not torch.is_tensor(self.baseline):
            if not torch.is_tensor(self.baseline):
                self.baseline_module = self.baseline
            state = self.final_state
            parts = [s.detach() for s in (state if isinstance(state, (tuple, list)) else [state])]
            self.baseline = self.baseline_module(self.obs, self.what.detach(), self.where.detach(),
                                                 self.presence.detach(), parts)            # sampled presence, :227
            self.baseline_vars = list(self.baseline_module.parameters())
        if decay_rate is not None:
            # EMA normalisation of the [B,B] importance weight (model.py:232-239, ops.py:46-64).  Off in the reference
            # script; evaluated with torch glue on the generic path (the fused air_nvil kernel covers decay_rate=None).
            iw = importance_weight.detach()[None, :] - self.baseline                      # [B,B]: (i,j) = imp_j - b_i
            mean, var = iw.detach().mean(), iw.detach().var(unbiased=False)
            # the EMA variables belong to THIS model and their update op runs only inside a train step (ops.py:46-64:
            # UPDATE_OPS are control dependencies of apply_gradients, model.py:357-359); evaluate() only reads them
            store, upd = self._moving_averages, self._in_train_step
            self.imp_weight_moving_mean = make_moving_average('imp_weight_moving_mean', mean, 0., decay_rate,
                                                              store=store, update=upd)
            self.imp_weight_moving_var = make_moving_average('imp_weight_moving_var', var, 1., decay_rate,
                                                             store=store, update=upd)
            factor = torch.clamp(torch.sqrt(self.imp_weight_moving_var), min=1.)
            iwn = (iw - self.imp_weight_moving_mean) / factor
            self.importance_weight = iwn.detach()
            self.imp_weight_mean, self.imp_weight_var = iwn.detach().mean(), iwn.detach().var(unbiased=False)
            self.reinforce_loss = (iwn.detach() * log_prob[None, :]).mean()
            self.baseline_loss = 0.5 * ((importance_weight.detach()[None, :] - self.baseline) ** 2).mean()
            return self.reinforce_loss
        # [B] - [B,1] -> [B,B] broadcast of the reference (SURVEY B-1), evaluated in closed form by air_nvil
        rl, bl, m, v = F.nvil(importance_weight.detach(), self.baseline, log_prob)
        self.importance_weight = importance_weight.detach()[None, :] - self.baseline.detach()
        self.imp_weight_mean, self.imp_weight_var = m, v
        self.reinforce_loss = rl
        self.baseline_loss = bl
        return self.reinforce_loss

    # ---- model.py:261-376 ------------------------------------------------------------------------------------------
    def _losses(self, global_step):
        loss = Loss()
        self.rec_loss_per_sample = F.rec_loglik(self.obs, self._canvas_unscaled[-1], float(self.output_multiplier),
                                                self.output_std)
        self.rec_loss = self.rec_loss_per_sample.mean()
        loss.add(self.rec_loss, self.rec_loss_per_sample)
        self.prior_loss = self._prior_loss(self.what_prior, self.where_scale_prior, self.where_shift_prior,
                                           self.num_steps_prior, global_step)
        self.prior_weight = 1.0 if self.use_prior else 0.0
        loss.add(self.prior_loss, weight=self.prior_weight)
        opt_loss = loss.value
        if self.use_reinforce:
            self.reinforce_imp_weight = self.rec_loss_per_sample
            if not getattr(self.num_steps_prior, 'analytic', True):
                self.reinforce_imp_weight = self.reinforce_imp_weight + self.prior_loss.per_sample
            opt_loss = opt_loss + self._reinforce(self.reinforce_imp_weight, self._decay_rate)
        if self.l2_weight and self.l2_weight > 0.:                                            # model.py:346-353
            bset = {id(p) for p in getattr(self, 'baseline_vars', [])}
            weights = [p for p in self.cell.parameters() if p.dim() == 2 and id(p) not in bset]   # incl. the [1,H] initial state, like len(shape)==2 in the reference
            self.l2_loss = self.l2_weight * sum((w * w).sum() / 2 for w in weights)
            opt_loss = opt_loss + self.l2_loss
        self.loss = loss
        self.opt_loss = opt_loss
        if self.nums is not None:
            self.num_step_accuracy = (self.gt_num_steps == self.num_step_per_sample).float().mean()
        return opt_loss

    def evaluate(self, obs=None, nums=None, noise=None):
        """Forward pass + every loss attribute on (obs, nums) WITHOUT an update -- what `sess.run(exprs, feed_dict)` did
        in the reference's logger (evaluation.py:137-164).  Requires `train_step(...)` to have been called (priors)."""
        with torch.no_grad():
            self.forward(obs, nums, noise)
            self._losses(self.global_step)
        return self

    def toggle_prior(self):
        self.use_prior = not self.use_prior
        return self.use_prior

    def train_step(self, learning_rate, l2_weight=0., what_prior=None, where_scale_prior=None,
                   where_shift_prior=None, num_steps_prior=None, use_prior=True, use_reinforce=True, baseline=None,
                   decay_rate=None, optimizer=None, opt_kwargs=None):
        """Creates the train step and the global_step (model.py:261-376).
        `optimizer` / `opt_kwargs` (model.py:265: `optimizer=tf.train.RMSPropOptimizer, opt_kwargs=dict(momentum=.9,
        centered=True)`, instantiated as optimizer(learning_rate, **opt_kwargs) and once more at 10x the rate for the
        baseline, model.py:355-363): the default runs on the HIP centred-RMSProp kernel with TF's slot initialisation; any
        other choice is given as a torch.optim class (same calling convention: optimizer(params, lr, **opt_kwargs)) and
        steps the generic autograd path."""
        custom_opt = optimizer is not None
        self._custom_optimizer = (optimizer, dict(opt_kwargs or {})) if custom_opt else None
        # the reference's default class with other keywords (model.py:265: RMSPropOptimizer(lr, **opt_kwargs), e.g. momentum=.5
        # or centered=False): tf.train.RMSPropOptimizer's own defaults for what is not given, all of them plain scalars of
        # the HIP update (air_rmsprop)
        rms_kw = dict(decay=0.9, momentum=0.0, epsilon=1e-10, centered=False)
        given = dict(momentum=.9, centered=True) if opt_kwargs is None else dict(opt_kwargs)
        if not custom_opt:
            unknown = set(given) - set(rms_kw) - {"use_locking", "name"}
            if unknown:
                raise TypeError("RMSPropOptimizer got unexpected keyword(s) %s" % sorted(unknown))
            rms_kw.update({k: v for k, v in given.items() if k in rms_kw})
        self._rms_kwargs = rms_kw
        self._default_rms = (not custom_opt and float(rms_kw["decay"]) == 0.9 and float(rms_kw["momentum"]) == 0.9
                             and float(rms_kw["epsilon"]) == 1e-10 and bool(rms_kw["centered"]))
        if num_steps_prior is not None and not hasattr(num_steps_prior, 'analytic'):
            num_steps_prior['analytic'] = True
        self.l2_weight = l2_weight
        self.what_prior, self.where_scale_prior = what_prior, where_scale_prior
        self.where_shift_prior, self.num_steps_prior = where_shift_prior, num_steps_prior
        if not hasattr(self, 'baseline'):
            self.baseline = baseline
        self.baseline_module = self.baseline if not torch.is_tensor(self.baseline) else None
        self.use_prior = bool(use_prior)
        self.use_reinforce = use_reinforce
        self._decay_rate = decay_rate
        self.learning_rate = torch.tensor(float(learning_rate))
        self.global_step = torch.zeros((), dtype=torch.int64)
        self._losses(self.global_step)                       # builds the baseline, exposes the loss attributes
        self._slots = {}
        lr_dev = torch.tensor([float(learning_rate)], device=self.obs.device)

        def _update(params, lr_mult):
            for p in params:
                if p.grad is None:
                    continue
                if p not in self._slots:
                    self._slots[p] = (torch.ones_like(p), torch.zeros_like(p), torch.zeros_like(p))
                ms, mg, mom = self._slots[p]
                H.rmsprop_centered_(p.data, p.grad, ms, mg, mom, lr_dev, lr_mult, decay=rms_kw["decay"],
                                    momentum=rms_kw["momentum"], eps=rms_kw["epsilon"], centered=rms_kw["centered"])

        def train_step_fn(obs=None, nums=None, noise=None):
            """One update (== sess.run(train_step)): fresh forward, both gradient sets, both RMSProp updates."""
            lr_dev.fill_(float(self.learning_rate))
            self.forward(obs, nums, noise)
            self._in_train_step = True
            try:
                opt_loss = self._losses(self.global_step)
            finally:
                self._in_train_step = False
            baseline_vars = list(getattr(self, 'baseline_vars', []))
            bset = {id(p) for p in baseline_vars}
            model_vars = [p for p in self.cell.parameters() if id(p) not in bset]
            for p in model_vars + baseline_vars:
                p.grad = None
            gm = torch.autograd.grad(opt_loss, model_vars, retain_graph=True, allow_unused=True)
            for p, g in zip(model_vars, gm):
                p.grad = g
            if self.use_reinforce and baseline_vars:
                gb = torch.autograd.grad(self.baseline_loss, baseline_vars, allow_unused=True)
                for p, g in zip(baseline_vars, gb):
                    p.grad = g
            if self._custom_optimizer is None:
                _update(model_vars, 1.0)
                _update(baseline_vars, 10.0)                  # model.py:363
            else:
                cls, kw = self._custom_optimizer
                if "model" not in self._slots:                # built on the first step: the baseline is created lazily
                    self._slots["model"] = cls(model_vars, lr=float(self.learning_rate), **kw)
                    if baseline_vars:
                        self._slots["baseline"] = cls(baseline_vars, lr=10.0 * float(self.learning_rate), **kw)
                for key, mult in (("model", 1.0), ("baseline", 10.0)):
                    if key in self._slots:
                        for grp in self._slots[key].param_groups:
                            grp["lr"] = mult * float(self.learning_rate)
                        self._slots[key].step()
            self.global_step += 1
            return self.global_step

        self._train_step = train_step_fn
        return self._train_step, self.global_step
