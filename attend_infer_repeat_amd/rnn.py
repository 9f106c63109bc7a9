"""Recurrent transitions with the Sonnet v1 RNNCore contract the reference's AIRCell expects from `transition`
(cell.py:43-44,78,103,127): `.output_size`, `.state_size`, `.initial_state(batch, dtype, trainable=)`, call ->
(output, new_state).  LSTM is what AIRonMNIST uses (mnist_model.py:35); GRU is what test/cell_test.py:12 uses.
All matmuls / gate math run in the HIP kernels.
"""
import torch

from . import functional as F
from . import hip as H
from .neural import _sonnet_linear_init


class LSTM(torch.nn.Module):
    """snt.LSTM: gates = [x,h].W + b, order i,j,f,o, forget bias 1; state (h, c)."""

    def __init__(self, hidden_size, forget_bias=1.0):
        super().__init__()
        self._hidden = int(hidden_size)
        self._forget_bias = float(forget_bias)
        self.w_gates = None
        self.b_gates = None
        self.h0 = None
        self.c0 = None

    @property
    def output_size(self):
        return (self._hidden,)

    @property
    def state_size(self):
        return ((self._hidden,), (self._hidden,))

    def initial_state(self, batch_size, dtype=torch.float32, trainable=False, device=None):
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.h0 is None:
            z = lambda: torch.zeros(1, self._hidden, device=device)
            if trainable:                                  # cell.py:103: trainable initial state
                self.h0, self.c0 = torch.nn.Parameter(z()), torch.nn.Parameter(z())
            else:
                self.register_buffer("h0", z()); self.register_buffer("c0", z())
        return (self.h0.expand(batch_size, -1).contiguous(), self.c0.expand(batch_size, -1).contiguous())

    def forward(self, x, state):
        h, c = state
        if self.w_gates is None:
            n_in = x.shape[-1] + self._hidden
            self.w_gates = torch.nn.Parameter(_sonnet_linear_init(n_in, 4 * self._hidden).to(x.device))
            self.b_gates = torch.nn.Parameter(torch.zeros(4 * self._hidden, device=x.device))
        h2, c2 = F.lstm_cell(x, h, c, self.w_gates, self.b_gates, self._forget_bias)
        return h2, (h2, c2)


class GRU(torch.nn.Module):
    """snt.GRU: z = sig(x.Wz + h.Uz + bz); r = sig(x.Wr + h.Ur + br); a = tanh(x.Wh + (r*h).Uh + bh);
    h' = (1-z)*h + z*a."""

    def __init__(self, hidden_size):
        super().__init__()
        self._hidden = int(hidden_size)
        self._built = False
        self.h0 = None

    @property
    def output_size(self):
        return (self._hidden,)

    @property
    def state_size(self):
        return (self._hidden,)

    def initial_state(self, batch_size, dtype=torch.float32, trainable=False, device=None):
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.h0 is None:
            z = torch.zeros(1, self._hidden, device=device)
            if trainable:
                self.h0 = torch.nn.Parameter(z)
            else:
                self.register_buffer("h0", z)
        return self.h0.expand(batch_size, -1).contiguous()

    def forward(self, x, h):
        if not self._built:
            n_in, Hd, dev = x.shape[-1], self._hidden, x.device
            for g in "zrh":
                setattr(self, f"w{g}", torch.nn.Parameter(_sonnet_linear_init(n_in, Hd).to(dev)))
                setattr(self, f"u{g}", torch.nn.Parameter(_sonnet_linear_init(Hd, Hd).to(dev)))
                setattr(self, f"b{g}", torch.nn.Parameter(torch.zeros(Hd, device=dev)))
            self._built = True
        lin = lambda a, w, b=None: F.linear(a, w, b, H.ACT_NONE)
        z = torch.sigmoid(lin(x, self.wz, self.bz) + lin(h, self.uz))
        r = torch.sigmoid(lin(x, self.wr, self.br) + lin(h, self.ur))
        a = torch.tanh(lin(x, self.wh, self.bh) + lin(r * h, self.uh))
        h2 = (1 - z) * h + z * a
        return h2, h2
