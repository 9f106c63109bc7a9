"""The oracle's restatements against PyTorch's own, independently written implementations of the same algorithms (CPU).

These do not pin TF 1.1 / Sonnet 1.1 conventions (gate ORDER, forget-bias placement, eps placement are assumptions listed in
tests/golden/ASSUMPTIONS.md); they pin the algebra: once the convention is mapped, a third party's LSTM cell and centred
RMSProp give the same numbers as oracle/air_oracle.py."""
import numpy as np
import torch

from oracle import air_oracle as O


def test_lstm_step_equals_torch_lstmcell_after_gate_permutation():
    torch.manual_seed(0)
    M, I, H = 6, 10, 7
    x, h, c = torch.randn(M, I).double(), torch.randn(M, H).double(), torch.randn(M, H).double()
    w = (torch.randn(I + H, 4 * H) / 4).double()              # Sonnet layout [in+hid, 4*hid], gate order i, j, f, o
    b = torch.randn(4 * H).double()
    h2, c2 = O.lstm_step(x, h, c, w, b, forget_bias=1.0)
    cell = torch.nn.LSTMCell(I, H).double()                    # torch: weight_ih [4H, I], gate order i, f, g, o
    i_, j_, f_, o_ = (slice(k * H, (k + 1) * H) for k in range(4))
    order = [i_, f_, j_, o_]
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([w[:I, s] for s in order], 1).t())
        cell.weight_hh.copy_(torch.cat([w[I:, s] for s in order], 1).t())
        bias = torch.cat([b[s] for s in order]).clone()
        bias[H:2 * H] += 1.0                                   # Sonnet adds forget_bias=1 inside the sigmoid
        cell.bias_ih.copy_(bias)
        cell.bias_hh.zero_()
    th, tc = cell(x, (h, c))
    assert torch.allclose(h2, th, rtol=1e-12, atol=1e-12) and torch.allclose(c2, tc, rtol=1e-12, atol=1e-12)


def test_centered_rmsprop_equals_torch_rmsprop_up_to_eps_placement():
    """TF: mom <- m*mom + lr*g/sqrt(ms - mg^2 + eps), ms initialised to ONE.  torch: buf <- m*buf + g/(sqrt(ms - mg^2) + eps),
    p <- p - lr*buf, ms initialised to zero.  With a constant lr and ms preset to one the trajectories differ only by the
    placement of eps = 1e-10 (relative 1e-10)."""
    torch.manual_seed(1)
    cfg = O.AIRConfig()
    p0 = torch.randn(5, 4).double()
    params = {"input_encoder/0/w": p0.clone()}                 # a model (not baseline) variable: lr multiplier 1
    slots = O.rmsprop_init(params)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([tp], lr=cfg.learning_rate, alpha=cfg.rms_decay, eps=cfg.rms_eps, momentum=cfg.rms_momentum,
                              centered=True)
    for it in range(5):
        g = torch.randn(5, 4, generator=torch.Generator().manual_seed(10 + it)).double()
        O.rmsprop_centered_step(params, {"input_encoder/0/w": g}, slots, cfg)
        tp.grad = g.clone()
        if it == 0:                                            # materialise torch's state, then apply TF's ms = 1 initialisation
            opt.step()
            st = opt.state[tp]
            with torch.no_grad():
                tp.copy_(p0); st["square_avg"].fill_(1.0); st["grad_avg"].zero_(); st["momentum_buffer"].zero_()
                st["step"] = torch.tensor(0.0) if torch.is_tensor(st["step"]) else 0
            tp.grad = g.clone()
        opt.step()
    np.testing.assert_allclose(params["input_encoder/0/w"].numpy(), tp.detach().numpy(), rtol=1e-8, atol=1e-12)
