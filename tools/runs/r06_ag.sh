mkdir -p gpurun_out/r06_ag
timeout 1500 python -m pytest tests -x -q -m gpu --durations=40 > gpurun_out/r06_ag/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06_ag/gpu_tests.log | tail -1
grep -A45 "slowest" gpurun_out/r06_ag/gpu_tests.log | head -48
