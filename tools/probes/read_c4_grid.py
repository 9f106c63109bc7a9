#!/usr/bin/env python
"""Round 6: the stand-alone glimpse read at configs[3] shapes (100x100 / 28x28 / T=5) out of cache, under AIR_ST_READ_GRID / AIR_ST_READ_THREADS
(both read once per process: run one setting per invocation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from attend_infer_repeat_amd.engine import EngineConfig
cfg = EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5)
r = bench.st_read_sweep(cfg, 5, [1024, 8192, 32768, 65536], torch.device("cuda", 0))
print("grid", os.environ.get("AIR_ST_READ_GRID", "-"), "threads", os.environ.get("AIR_ST_READ_THREADS", "-"), [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
