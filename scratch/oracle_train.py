import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from oracle import air_oracle as O
torch.set_num_threads(8)
cfg = O.AIRConfig()
params = O.init_params(cfg, seed=1)
slots = O.rmsprop_init(params)
data, _ = O.synthetic_batch(cfg, 4096, seed=0)
rng = np.random.default_rng(0)
t0 = time.time()
for it in range(6001):
    idx = torch.tensor(rng.integers(0, 4096, 64))
    obs = data[idx]
    res, _ = O.train_step(params, slots, cfg, obs, O.make_noise(cfg, 64, seed=1000 + it), global_step=it)
    if it % 500 == 0:
        print(it, "loss %.2f rec %.2f kl_n %.2f kl_what %.3f kl_where %.3f num_step %.2f  t=%.0fs" % (
            res["loss"].item(), res["rec_loss"].item(), res["kl_num_steps"].item(), res["kl_what"].item(),
            res["kl_where"].item(), res["num_step"].item(), time.time() - t0), flush=True)
