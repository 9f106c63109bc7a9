import sys, torch
sys.path.insert(0, '.')
from oracle import air_oracle as O
from tests.test_engine import make_pair
for lr, steps in ((1e-4, 300), (3e-4, 150)):
    ocfg, B = O.AIRConfig(learning_rate=lr), 32
    eng, params, obs, noise = make_pair(ocfg, B, bias_std=0.0)
    eng.set_learning_rate(lr)
    eng.forward(); first = eng.outputs()["loss"].item()
    eng.capture()
    hist = []
    for i in range(steps):
        eng.train_step()
        if i % (steps // 6) == 0:
            eng.synchronize(); hist.append(round(eng.outputs()["loss"].item(), 1))
    eng.forward(); last = eng.outputs()["loss"].item()
    print(lr, steps, "first", first, "last", last, "finite", bool(torch.isfinite(eng.flat_params).all()), hist)
