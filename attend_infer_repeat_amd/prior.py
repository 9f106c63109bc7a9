"""Number-of-steps distribution math with the interface of the reference's prior.py (attend_infer_repeat/prior.py).

These are the general-shape, device-agnostic utilities (float64 internally, like the reference) that the reference's
own tests exercise (test/prior_test.py); the training hot path evaluates the same math in the HIP kernels
air_numsteps_fwd/_bwd and air_steps_prior.  `NumStepsDistribution` dispatches to the kernels for [B,T] HIP tensors.
"""
import numpy as np
import torch

from .ops import clip_preserve


def masked_apply(tensor, op, mask):
    """op(tensor) where mask, 0 elsewhere, NaN/inf-safe in value and gradient (prior.py:8-23)."""
    safe = torch.where(mask, tensor, torch.ones_like(tensor))
    return torch.where(mask, op(safe), torch.zeros_like(tensor))


def geometric_prior(success_prob, n_steps):
    """p(n) = (1-s) s^n for n = 0..n_steps with s clipped to [1e-7, 1-1e-15]; NOT renormalised (prior.py:26-32).
    Python floats / float64 tensors evaluate in float64 (the annealed path), float32 tensors in float32."""
    if torch.is_tensor(success_prob):
        s = success_prob
        dtype = s.dtype if s.dtype in (torch.float32, torch.float64) else torch.float64
    else:
        s, dtype = torch.tensor(float(success_prob), dtype=torch.float64), torch.float64
    s = s.to(dtype).clamp(1e-7, 1.0 - 1e-15)
    probs = 1.0 - s
    k = torch.arange(n_steps + 1, dtype=dtype, device=s.device)
    return torch.exp(k * torch.log1p(-probs) + torch.log(probs))


def _cumprod(tensor, axis=0):
    """cumprod whose gradient is NaN-free at zeros (prior.py:35-59): products, never divisions."""
    n = tensor.shape[axis]
    parts, acc = [], None
    for i in range(n):
        x = tensor.select(axis, i)
        acc = x if acc is None else acc * x
        parts.append(acc)
    return torch.stack(parts, axis)


def bernoulli_to_modified_geometric(presence_prob):
    """q(n) = [1-p1, p1(1-p2), ..., prod p] in float64, renormalised, cast to float32 (prior.py:62-68)."""
    presence_prob = presence_prob.to(torch.float64)
    inv = 1. - presence_prob
    prob = _cumprod(presence_prob, axis=-1)
    modified_prob = torch.cat([inv[..., :1], inv[..., 1:] * prob[..., :-1], prob[..., -1:]], -1)
    modified_prob = modified_prob / modified_prob.sum(-1, keepdim=True)
    return modified_prob.to(torch.float32)


def tabular_kl(p, q, zero_prob_value=0., logarg_clip=None):
    """Per-coordinate p*log(p/q) for tabular pmfs, 0 where p <= zero_prob_value (prior.py:71-90); f64 -> f32."""
    p, q = (torch.as_tensor(i).to(torch.float64) for i in (p, q))
    non_zero = p > zero_prob_value
    logarg = p / q
    if logarg_clip is not None:
        logarg = clip_preserve(logarg, 1. / logarg_clip, logarg_clip)
    log = masked_apply(logarg, torch.log, non_zero)
    return (p * log).to(torch.float32)


def sample_from_1d_tensor(arr, idx):
    arr = torch.as_tensor(arr)
    assert arr.dim() == 1, "shape is {}".format(tuple(arr.shape))
    return arr[idx.to(torch.int64)]


def sample_from_tensor(tensor, idx):
    """tensor[b, idx[b]] for minibatches (prior.py:103-116)."""
    tensor = torch.as_tensor(tensor)
    if tensor.dim() > 2:
        raise NotImplementedError
    idx = idx.to(torch.int64)
    shift = torch.arange(tensor.shape[0], device=tensor.device) * tensor.shape[1]
    p_flat = tensor.reshape(-1)
    idx_flat = idx.reshape(-1) + shift
    return sample_from_1d_tensor(p_flat, idx_flat).reshape(idx.shape)


class NumStepsDistribution(object):
    """Bernoulli step probabilities -> p(n) over the number of steps (prior.py:119-151)."""

    def __init__(self, steps_probs):
        self._steps_probs = steps_probs
        self._joint = bernoulli_to_modified_geometric(steps_probs)

    def sample(self, n=None):
        shape = tuple(self._steps_probs.shape) if n is None else (n,) + tuple(self._steps_probs.shape)
        sample = (torch.rand(shape, device=self._steps_probs.device) < self._steps_probs).to(torch.float32)
        return torch.cumprod(sample, -1).sum(-1)

    def prob(self, samples=None):
        if samples is None:
            return self._joint
        return sample_from_tensor(self._joint, samples)

    def log_prob(self, samples):
        prob = self.prob(samples)
        prob = clip_preserve(prob, 1e-32, prob.detach())
        return torch.log(prob)
