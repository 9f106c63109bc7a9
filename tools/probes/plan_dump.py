"""Developer probe: the train step's launch list with the problems of every grouped GEMM (run on the GPU box).
   python tools/probes/plan_dump.py [c2|c4|c5] [batch]"""
import sys, ctypes
import torch
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
from attend_infer_repeat_amd import _lib

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kw = dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5) if cfgname == "c4" else {}
if cfgname == "c5":
    kw, B = dict(mfma_dtype="bf16"), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
eng = AIREngine(EngineConfig(**kw), B, device=torch.device("cuda", 0), seed=1, keep_canvas_steps=True)
plans = eng._single_gpu_step_plans()
i = 0
for plan in plans:
    for fn, args, name in plan:
        line = "%2d %-34s" % (i, name)
        if name in ("air_gemm", "air_gemm_bf16"):
            line += "ta=%d tb=%d %dx%dx%d epi=%s" % (args[0], args[1], args[2], args[3], args[4], args[12])
        if name.startswith("air_gemm_grouped"):
            arr, n = args[0], args[1]
            probs = []
            for j in range(n):
                d = arr[j]
                t16 = ((d.M + 15) // 16) * ((d.N + 15) // 16)
                probs.append("%s%s %dx%dx%d(%d)%s" % ("T" if d.ta else "N", "T" if d.tb else "N", d.M, d.N, d.K, t16, "+cs" if d.colsum else ""))
            line += " | ".join(probs)
        print(line)
        i += 1
