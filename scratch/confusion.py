"""train the script's model for N iterations with a seed, then print the (gt count, predicted steps) confusion matrix"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from attend_infer_repeat_amd.scripts import multi_mnist as S

seed, iters = int(sys.argv[1]), int(sys.argv[2])
air = S.main(["--iters", str(iters), "--log-every", sys.argv[3] if len(sys.argv) > 3 else str(10 ** 9), "--save-every", str(10 ** 9), "--results-dir",
              "gpurun_out/confusion", "--run-name", "s%d" % seed, "--seed", str(seed)])
from attend_infer_repeat_amd.data import synthetic_dataset, DeviceFeeder
valid = synthetic_dataset(10000, seed=seed + 1)
feed = DeviceFeeder(valid, 64, torch.device("cuda", 0), shuffle=False)
conf = np.zeros((3, 4), int)
pp = []
for _ in range(50):
    x, y = feed()
    air.evaluate(x, y)
    gt = air.gt_num_steps.cpu().numpy().astype(int)
    pr = air.num_step_per_sample.cpu().numpy().astype(int)
    for g, p in zip(gt, pr):
        conf[g, p] += 1
    pp.append(air.presence_prob.reshape(3, -1).mean(1).cpu().numpy())
print("rows = gt objects 0..2, cols = predicted steps 0..3")
print(conf)
print("accuracy", np.trace(conf[:, :3]) / conf.sum(), " mean presence_prob per step", np.mean(pp, 0))
