#!/bin/bash
O=gpurun_out/r04_n; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_golden.py -q -m gpu -k "st_read or golden or read" > $O/read_tests.log 2>&1; echo "read tests rc=$?"; tail -3 $O/read_tests.log
for TH in 0 256 320 384 512; do
AIR_ST_READ_THREADS=$TH python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import st_read_sweep
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda:0")
for cfg, T in ((EngineConfig(), 3), (EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5)):
    r = st_read_sweep(cfg, T, [64, 8192, 65536], dev)
    print("threads", os.environ["AIR_ST_READ_THREADS"], cfg.img_size, "T", T, [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
PY
done 2>&1 | grep -v amdgpu.ids | tee $O/read_threads.txt
