#!/bin/bash
# round 4, call m: the lean ST glimpse read: parity + out-of-cache sweep against the pipelined kernel (same box)
O=gpurun_out/r04_m; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_golden.py -q -m gpu -k "st_read or golden or read" > $O/read_tests.log 2>&1; echo "read tests rc=$?"; tail -5 $O/read_tests.log
for L in 1 2; do
AIR_ST_READ_LEAN=$L python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from bench import st_read_sweep
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda:0")
for cfg, T in ((EngineConfig(), 3), (EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5)):
    r = st_read_sweep(cfg, T, [64, 1024, 8192, 65536], dev)
    print("lean", os.environ["AIR_ST_READ_LEAN"], cfg.img_size, "T", T, [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
    r = st_read_sweep(cfg, 1, [3072, 24576, 196608] if T == 3 else [3072, 24576], dev, share_image=False)
    print("lean", os.environ["AIR_ST_READ_LEAN"], cfg.img_size, "1:1", [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
PY
done 2>&1 | grep -v amdgpu.ids | tee $O/read_sweep.txt
