import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import air_oracle as O
from tests.test_engine import make_pair
ocfg, B = O.AIRConfig(learning_rate=1e-3), 32
eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0)
eng.set_learning_rate(1e-3)
for it in range(80):
    eng.train_step(); eng.synchronize()
    g = eng.flat_grads
    if not torch.isfinite(g).all():
        dw = eng.dwhere_w.view(-1, 4)
        bad = (~torch.isfinite(dw)).any(1).nonzero().flatten()
        print("iter", it, "bad rows", bad.tolist()[:10])
        dec = eng.gd.out[-1]
        print("decoded finite", torch.isfinite(dec).all().item(), dec.abs().max().item())
        print("final canvas finite", torch.isfinite(eng.final_canvas).all().item(), eng.final_canvas.abs().max().item())
        for r in bad.tolist()[:4]:
            print("row", r, "where", eng.where.view(-1, 4)[r].tolist(), "pres", eng.presence.view(-1)[r].item(),
                  "dwhere_w", dw[r].tolist(), "dec max", dec[r].abs().max().item(), "dg max", eng.gd.g[-1][r].abs().max().item())
        break
