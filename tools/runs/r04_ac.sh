#!/bin/bash
# resident-multiple grids everywhere: parity of the touched kernels, A/B against the previous build, the bench sweeps, configs[4]
O=gpurun_out/r04_ac; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_golden.py tests/test_extreme_scales.py -q -m gpu -x -k "canvas or write or golden or degenerate or extreme or read" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
python tools/probes/read_bwd_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/read_bwd_ab.txt
READ_AB_FULL=1 python tools/probes/read_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/read_ab.txt
python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench.log
python bench.py --config c5 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > $O/bench_c5.json 2>> $O/bench.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_ac/bench_c2.json").read().strip().splitlines()[-1])
print("c2", d["value"], d["ms_per_step"])
for k, v in d.items():
    if k.startswith("roofline_sweep"):
        print(k, [(e.get("batch"), e.get("us_per_launch"), e.get("frac")) for e in v] if isinstance(v, list) else v)
d = json.loads(open("gpurun_out/r04_ac/bench_c5.json").read().strip().splitlines()[-1])
print("c5", d["value"], d["ms_per_step"])
PY
