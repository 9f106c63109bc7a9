import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import air_oracle as O
from tests.test_engine import make_pair
ocfg, B = O.AIRConfig(learning_rate=1e-3), 32
eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0)
eng.set_learning_rate(1e-3)
names_f = ["where", "where_scale", "glimpse_in", "what", "what_scale", "final_canvas", "rec", "kl_n", "logp", "step_w", "kl_what_row", "kl_where_row", "nvil_out", "dlogp", "dbase"]
names_b = ["dwhere_w", "d_what", "dq", "d_glimpse_in", "dwhere_r", "dH", "dH_b", "dgates", "dgx", "dh_init"]
for it in range(80):
    prev = eng.flat_params.clone()
    eng.train_step(); eng.synchronize()
    if not torch.isfinite(eng.flat_grads).all() or not torch.isfinite(eng.flat_params).all():
        print("non-finite at iter", it, "loss", eng.outputs()["loss"].item())
        for nm in names_f + names_b:
            t = getattr(eng, nm)
            print(f"  {nm:14s} finite={bool(torch.isfinite(t).all())} absmax={t[torch.isfinite(t)].abs().max().item() if torch.isfinite(t).any() else float('nan'):.3e}")
        for nm, m in (("gd", eng.gd), ("ge", eng.ge), ("tr", eng.tr), ("st", eng.st), ("enc", eng.enc), ("bl", eng.bl)):
            for i, (o, g) in enumerate(zip(m.out, m.g)):
                print(f"  {nm}.out{i} finite={bool(torch.isfinite(o).all())} max={o.abs().max().item():.3e} | g{i} finite={bool(torch.isfinite(g).all())}")
        w = eng.where.view(-1, 4); ws = eng.where_scale.view(-1, 4)
        print("  min |sx| %.3e min |sy| %.3e  min where_scale %.3e" % (w[:, 0].abs().min().item(), w[:, 2].abs().min().item(), ws.min().item()))
        bad = (~torch.isfinite(eng.dwhere_w.view(-1, 4))).any(1).nonzero().flatten().tolist()
        for r in bad[:5]:
            print("  row", r, "where", w[r].tolist(), "pres", eng.presence.view(-1)[r].item(), "dwhere_w", eng.dwhere_w.view(-1, 4)[r].tolist())
        print("  params prev finite", bool(torch.isfinite(prev).all()), "prev absmax %.3e" % prev.abs().max().item())
        break
else:
    print("no NaN in 80 iters; loss", eng.outputs()["loss"].item())
