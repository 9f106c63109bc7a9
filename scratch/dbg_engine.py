import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import air_oracle as O
from tests.test_engine import make_pair
ocfg, B = O.AIRConfig(learning_rate=1e-3), 32
for mode in ["eager", "graph"]:
    eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0)
    eng.set_learning_rate(float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
    if mode == "graph":
        eng.capture()
    for it in range(60):
        eng.train_step(); eng.synchronize()
        o = eng.outputs()
        g = eng.flat_grads
        if it % 5 == 0 or not torch.isfinite(g).all():
            print(mode, it, "loss %.3f rec %.3f klw %.3f gmax %.3e gnan %d pmax %.3f min|sx| %.2e min ws %.2e nvil %s" % (
                o["loss"].item(), o["rec_loss"].item(), o["kl_what"].item(), g.abs().max().item(),
                (~torch.isfinite(g)).sum().item(), eng.flat_params.abs().max().item(),
                eng.where[..., 0].abs().min().item(), eng.what_scale.min().item(), eng.nvil_out.tolist()))
        if not torch.isfinite(g).all():
            bad = [k for k, v in eng.grads.items() if not torch.isfinite(v).all()]
            print("bad grads:", bad[:8])
            for nm in ["dwhere_w", "dwhere_r", "dprob", "d_what", "dq", "dH", "dgates"]:
                t = getattr(eng, nm); print(nm, torch.isfinite(t).all().item(), t.abs().max().item())
            break
