"""MLP / Affine building blocks with the interface of the reference's neural.py, computed by the fused HIP
GEMM+bias+ELU kernel (air_linear_fwd / air_linear_bwd).

Reference: attend_infer_repeat/neural.py:42-102.  Modules are lazily built on first call (Sonnet style: the input
width is taken from the first input).  Weights use Sonnet's layout w[in, out]; default init is Sonnet's
TruncNormal(0, 1/sqrt(fan_in)) with zero bias -- the reference's custom `default_init` never takes effect because
Affine passes it positionally into snt.Linear's `use_bias` slot (neural.py:53).
"""
import math

import torch

from . import functional as F
from . import hip as H


def elu(x):
    return torch.nn.functional.elu(x)


def selu(x):                                              # neural.py:13-17 (unused by the model)
    alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
    return scale * torch.where(x >= 0.0, x, alpha * torch.nn.functional.elu(x))


default_activation = elu


def create_linear_initializer(input_size):               # neural.py:7-10 (dead in the reference, kept for the surface)
    stddev = 1.0 / math.sqrt(input_size * 2)
    return lambda shape: torch.nn.init.trunc_normal_(torch.empty(shape), std=stddev, a=-2 * stddev, b=2 * stddev)


default_init = {"w": create_linear_initializer, "b": torch.zeros}


def activation_based_init(nonlinearity):                  # neural.py:28-39 (dead in the reference)
    def init(shape):
        fan_in = shape[0]
        factor = 1.0 if nonlinearity is selu else 2.0
        return torch.randn(shape) * math.sqrt(factor / fan_in)
    return init


def _sonnet_linear_init(n_in, n_out, generator=None):
    std = 1.0 / math.sqrt(n_in)
    w = torch.empty(n_in, n_out)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=generator)
    return w


class Affine(torch.nn.Module):
    """transfer(x.W + b)   (neural.py:42-60).  `transfer` elu/None are fused into the GEMM epilogue; any other callable
    is applied after a linear kernel."""

    def __init__(self, n_output, transfer=default_activation, initializers=None, transfer_based_init=False):
        super().__init__()
        self._n_output = int(n_output)
        self._transfer = transfer
        self.w = None
        self.b = None

    def _build_params(self, n_in, device):
        self.w = torch.nn.Parameter(_sonnet_linear_init(n_in, self._n_output).to(device))
        self.b = torch.nn.Parameter(torch.zeros(self._n_output, device=device))

    @property
    def output_size(self):
        return self._n_output

    def forward(self, inpt):
        if self.w is None:
            self._build_params(inpt.shape[-1], inpt.device)
        if self._transfer is None:
            return F.linear(inpt, self.w, self.b, H.ACT_NONE)
        if self._transfer is elu or self._transfer is torch.nn.functional.elu:
            return F.linear(inpt, self.w, self.b, H.ACT_ELU)
        return self._transfer(F.linear(inpt, self.w, self.b, H.ACT_NONE))


def _flatten(x):
    if isinstance(x, (list, tuple)):
        out = []
        for i in x:
            out.extend(_flatten(i))
        return out
    return [x]


class MLP(torch.nn.Module):
    """Stack of Affines (neural.py:63-102): hidden layers with `hidden_transfer`, optional output layer `n_out` with
    `transfer` (None = linear)."""

    def __init__(self, n_hiddens, hidden_transfer=default_activation, n_out=None, transfer=None,
                 initializers=default_init):
        super().__init__()
        self._n_hiddens = [int(n) for n in _flatten(n_hiddens)]
        transfers = _flatten(hidden_transfer)
        if len(transfers) > 1:
            assert len(transfers) == len(self._n_hiddens)
        else:
            transfers = transfers * len(self._n_hiddens)
        self._hidden_transfers = transfers
        self._n_out = n_out
        self._transfer = transfer
        layers = [Affine(n, t) for n, t in zip(self._n_hiddens, self._hidden_transfers)]
        if n_out is not None:
            layers.append(Affine(n_out, transfer))
        self.layers = torch.nn.ModuleList(layers)

    @property
    def output_size(self):
        return self._n_out if self._n_out is not None else self._n_hiddens[-1]

    def forward(self, inpt):
        for layer in self.layers:
            inpt = layer(inpt)
        return inpt
