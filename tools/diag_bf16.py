"""diagnostic: bf16-operand engine vs bf16-emulating oracle, per-tensor errors at several batch sizes"""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import air_oracle as O
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig

def rel(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
def l2(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

for B in [int(x) for x in sys.argv[1:]] or [64, 704, 1024]:
    ocfg = O.AIRConfig()
    fields = {f.name for f in dataclasses.fields(EngineConfig)}
    ecfg = EngineConfig(mfma_dtype="bf16", **{k: v for k, v in dataclasses.asdict(ocfg).items() if k in fields})
    eng = AIREngine(ecfg, B, seed=1)
    params = O.init_params(ocfg, seed=1, bias_std=0.1)
    eng.load_parameters(params)
    obs, _ = O.synthetic_batch(ocfg, B, seed=11); noise = O.make_noise(ocfg, B, seed=21)
    eng.set_obs(obs.cuda()); eng.set_noise(noise["eps_where"].cuda(), noise["eps_what"].cuda(), noise["u_pres"].cuda())
    eng.set_global_step(20000)
    eng.forward(sample_noise=False); eng.backward()
    out, g = eng.outputs(), eng.named_grads()
    with O.matmul_mode("bf16"):
        res, grads = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    res32, grads32 = O.forward_backward(params, ocfg, obs, noise, global_step=20000)
    print("B", B, "presence equal", torch.equal(out["presence"].cpu().reshape(-1), res["presence"].reshape(-1)))
    for k in sorted(grads):
        print("  %-22s max %.2e l2 %.2e   | vs fp32 oracle: max %.2e l2 %.2e | bf16 oracle vs fp32 oracle l2 %.2e" % (
            k, rel(g[k], grads[k]), l2(g[k], grads[k]), rel(g[k], grads32[k]), l2(g[k], grads32[k]), l2(grads[k], grads32[k])))
