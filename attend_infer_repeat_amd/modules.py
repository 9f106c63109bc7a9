"""The AIR sub-networks with the interface of the reference's modules.py (attend_infer_repeat/modules.py:11-143),
computed by the HIP kernels: fused GEMM layers, the fused affine-grid + bilinear spatial transformer, the
reparameterised-Gaussian kernel.
"""
import numpy as np
import torch

from . import functional as F
from .neural import MLP
from .neural import _sonnet_linear_init
from . import hip as H


class NormalWithSoftplusScale(object):
    """tf.contrib.distributions.NormalWithSoftplusScale as used at cell.py:130-133,154-156: loc, scale=softplus(raw),
    reparameterised sample.  Built from the fused kernel's pre-activation block [loc_pre | raw]."""

    def __init__(self, loc_pre, raw_scale, raw_offset=0.0, loc_mode=0, guard_eps=0.0):
        self._pre = torch.cat([loc_pre, raw_scale], -1)
        self._raw_offset, self._loc_mode = raw_offset, loc_mode
        self._guard_eps = float(guard_eps or 0.0)            # the optional stability switch (AIRCell(guard_degenerate=...)); 0 = off
        self._evaluated = None

    def _eval(self, eps=None):
        D = self._pre.shape[-1] // 2
        if eps is None:
            eps = torch.randn(self._pre.shape[:-1] + (D,), device=self._pre.device)
        self._evaluated = F.gauss_sample(self._pre, eps, self._raw_offset, self._loc_mode, self._guard_eps)
        return self._evaluated

    def _get(self, i):
        if self._evaluated is None:
            self._eval()
        return self._evaluated[i]

    @property
    def loc(self):
        return self._get(0)

    @property
    def scale(self):
        return self._get(1)

    def sample(self, eps=None):
        if eps is not None or self._evaluated is None:
            self._eval(eps)
        return self._evaluated[2]


class ParametrisedGaussian(torch.nn.Module):
    """Linear(2n) -> NormalWithSoftplusScale(loc, raw + scale_offset)   (modules.py:11-24)."""

    def __init__(self, n_params, scale_offset=0., *args, **kwargs):
        super().__init__()
        self._n_params = int(n_params)
        self._scale_offset = float(scale_offset)
        self.guard_eps = 0.0
        self.w = None
        self.b = None

    def forward(self, inpt):
        if self.w is None:
            self.w = torch.nn.Parameter(_sonnet_linear_init(inpt.shape[-1], 2 * self._n_params).to(inpt.device))
            self.b = torch.nn.Parameter(torch.zeros(2 * self._n_params, device=inpt.device))
        params = F.linear(inpt, self.w, self.b, H.ACT_NONE)
        n = self._n_params
        return NormalWithSoftplusScale(params[..., :n], params[..., n:], raw_offset=self._scale_offset, loc_mode=0,
                                       guard_eps=self.guard_eps)


class TransformParam(torch.nn.Module):
    """modules.py:27-49.  NB the reference's `_build` calls itself (infinite recursion, SURVEY B-4); only the
    stochastic subclass is usable there.  Here `forward` does what was evidently meant: embed, then transform."""

    def __init__(self, n_hidden, n_param, max_crop_size=1.0):
        super().__init__()
        self._n_hidden = n_hidden
        self._n_param = int(n_param)
        self._max_crop_size = max_crop_size
        self.mlp = MLP(self._n_hidden, n_out=self._n_param)

    def _embed(self, inpt):
        return self.mlp(inpt.reshape(inpt.shape[0], -1))

    def _transform(self, inpt):
        sx, tx, sy, ty = torch.split(inpt, 1, 1)
        sx, sy = (self._max_crop_size * torch.sigmoid(s) for s in (sx, sy))
        tx, ty = (torch.tanh(t) for t in (tx, ty))
        return torch.cat((sx, tx, sy, ty), -1)

    def forward(self, inpt):
        return self._transform(self._embed(inpt))


class StochasticTransformParam(TransformParam):
    """Returns (locs, raw scales + scale_bias); locs = [sigmoid, tanh, sigmoid, tanh] (modules.py:52-63).  The loc
    squashing is fused into the Gaussian kernel downstream: this module returns the *pre-activation* locs tagged so
    that AIRCell can hand them to the kernel; called stand-alone it applies `_transform` itself."""

    def __init__(self, n_hidden, n_param, max_crop_size=1.0, scale_bias=-2.):
        super().__init__(n_hidden, n_param * 2, max_crop_size)
        self._scale_bias = scale_bias

    def embed_split(self, inpt):
        embedding = self._embed(inpt)
        n_params = self._n_param // 2                       # Py2 integer division at modules.py:60
        return embedding[..., :n_params], embedding[..., n_params:]

    @property
    def scale_bias(self):
        b = self._scale_bias
        return float(b.item()) if torch.is_tensor(b) else float(b)

    def forward(self, inpt):
        loc_pre, raw = self.embed_split(inpt)
        return self._transform(loc_pre), raw + self.scale_bias


class Encoder(torch.nn.Module):
    """BatchFlatten -> MLP(n_hidden)   (modules.py:66-76)."""

    def __init__(self, n_hidden):
        super().__init__()
        self._n_hidden = n_hidden
        self.mlp = MLP(n_hidden)

    def forward(self, inpt):
        return self.mlp(inpt.reshape(inpt.shape[0], -1))


class Decoder(torch.nn.Module):
    """MLP(n_hidden, n_out=prod(output_size)) -> BatchReshape(output_size)   (modules.py:79-91)."""

    def __init__(self, n_hidden, output_size):
        super().__init__()
        self._n_hidden = n_hidden
        self._output_size = tuple(int(s) for s in output_size)
        self.mlp = MLP(n_hidden, n_out=int(np.prod(self._output_size)))

    def forward(self, inpt):
        return self.mlp(inpt).reshape((inpt.shape[0],) + self._output_size)


class AffineWarpConstraints(object):
    """Stand-in for snt.AffineWarpConstraints: only the no-shear 2-D constraint the reference uses (cell.py:56)."""

    def __init__(self, kind="no_shear_2d"):
        self.kind = kind

    @classmethod
    def no_shear_2d(cls):
        return cls("no_shear_2d")


class SpatialTransformer(torch.nn.Module):
    """snt.AffineGridWarper(img_size, crop_size, no_shear_2d)[.inverse()] + snt.resampler, fused in one HIP kernel
    (modules.py:94-109).  inverse=False: glimpse read of size crop_size; inverse=True: a crop_size glimpse is warped
    back onto an img_size canvas.  transform_params rows are [sx, tx, sy, ty]."""

    def __init__(self, img_size, crop_size, constraints=None, inverse=False):
        super().__init__()
        if constraints is not None and getattr(constraints, "kind", "no_shear_2d") != "no_shear_2d":
            raise NotImplementedError("only AffineWarpConstraints.no_shear_2d() is supported (what AIR uses)")
        self._img_size = tuple(int(s) for s in img_size)
        self._crop_size = tuple(int(s) for s in crop_size)
        self._inverse = bool(inverse)

    def forward(self, img, transform_params):
        if img.dim() == 4:
            img = img[..., 0]
        if self._inverse:
            out = F.st_write(img, transform_params, self._img_size)
        else:
            out = F.st_read(img, transform_params, self._crop_size)
        return out[..., None]                                # resampler returns [B, h, w, C=1]


class StepsPredictor(torch.nn.Module):
    """sigmoid(MLP(n_hidden, n_out=1)(x) + steps_bias)   (modules.py:112-122)."""

    def __init__(self, n_hidden, steps_bias=0.):
        super().__init__()
        self._n_hidden = n_hidden
        self._steps_bias = steps_bias
        self.mlp = MLP(n_hidden, n_out=1)

    @property
    def steps_bias(self):
        b = self._steps_bias
        return float(b.item()) if torch.is_tensor(b) else float(b)

    def logit(self, inpt):
        return self.mlp(inpt)

    def forward(self, inpt):
        return torch.sigmoid(self.logit(inpt) + self.steps_bias)


class BaselineMLP(torch.nn.Module):
    """NVIL baseline (modules.py:125-143): MLP on [img | what | where | presence | state] -> [B,1]."""

    def __init__(self, n_hidden):
        super().__init__()
        self._n_hidden = n_hidden
        self.mlp = MLP(n_hidden, n_out=1)

    def forward(self, img, what, where, presence_prob, state=None):
        parts = []
        if state is not None:
            parts = [s for s in (state if isinstance(state, (tuple, list)) else [state])]
        inpt = H.baseline_pack(img.contiguous(), what.contiguous(), where.contiguous(), presence_prob.contiguous(),
                               [p.contiguous() for p in parts])
        return self.mlp(inpt)
