#!/bin/bash
# ST read, vectorised form with group descriptors and packed products: parity first, then the sweep (T glimpses per staged image, and 1:1)
O=gpurun_out/r04_t; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_golden.py tests/test_extreme_scales.py -q -m gpu -k "read or golden" > $O/read_tests.log 2>&1; echo "read tests rc=$?"; tail -3 $O/read_tests.log
for V in "2 16384" "2 8192" "2 4096" "1 16384"; do
set -- $V
AIR_ST_READ_VEC_MIN_R=$1 AIR_ST_READ_GRID=$2 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import st_read_sweep
from attend_infer_repeat_amd.engine import EngineConfig
dev = torch.device("cuda:0")
for cfg, T in ((EngineConfig(), 3), (EngineConfig(img_size=(100, 100), crop_size=(28, 28), max_steps=5), 5)):
    r = st_read_sweep(cfg, T, [64, 1024, 8192, 65536], dev)
    print("min_r", os.environ["AIR_ST_READ_VEC_MIN_R"], "grid", os.environ["AIR_ST_READ_GRID"], cfg.img_size, "T", T, [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
    if os.environ["AIR_ST_READ_GRID"] == "16384":
        r = st_read_sweep(cfg, 1, [3072, 24576] + ([196608] if T == 3 else []), dev)
        print("min_r", os.environ["AIR_ST_READ_VEC_MIN_R"], cfg.img_size, "1:1", [(x["batch"], x["us_per_launch"], x["frac"]) for x in r])
PY
done 2>&1 | grep -v amdgpu.ids | tee $O/read_vec.txt
