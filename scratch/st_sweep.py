import sys, json, torch
sys.path.insert(0, '.')
import bench
from attend_infer_repeat_amd.engine import EngineConfig
cfg = EngineConfig()
dev = torch.device('cuda', 0)
print("shared (T=3):")
for r in bench.st_read_sweep(cfg, 3, [64, 1024, 8192, 65536], dev): print(r)
print("one image per glimpse:")
for r in bench.st_read_sweep(cfg, 1, [192, 3072, 24576, 196608], dev, share_image=False): print(r)
cfg4 = EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5)
print("c4 100x100/28x28 one image per glimpse:")
for r in bench.st_read_sweep(cfg4, 1, [320, 8192, 65536], dev, share_image=False): print(r)
