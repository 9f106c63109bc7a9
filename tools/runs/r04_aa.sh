#!/bin/bash
# ST read out of cache: how many workgroups (grid-stride beyond) -- one image per workgroup pays the per-workgroup preamble per image
O=gpurun_out/r04_aa; mkdir -p $O
for G in 16384 8192 4096 2048 1024 16384; do
AIR_ST_READ_GRID=$G python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import st_read_sweep
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda:0")
for cfg, T in ((EngineConfig(), 3), (EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5)):
    r = st_read_sweep(cfg, T, [4096, 8192, 16384, 65536], dev)
    print("grid", os.environ["AIR_ST_READ_GRID"], cfg.img_size, "T", T, [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
PY
done 2>&1 | grep -v amdgpu.ids | tee $O/read_grid.txt
