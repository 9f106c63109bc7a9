OUT=gpurun_out/r06_f; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log; grep -E "^(FAILED|ERROR)" $OUT/gpu_tests.log | head -20
