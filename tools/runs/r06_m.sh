OUT=gpurun_out/r06_m; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_engine.py -q -m gpu -k "many_ranks and ipc" > $OUT/dp_tests.log 2>&1; tail -5 $OUT/dp_tests.log; grep -E "^(FAILED|ERROR)|Error|error" $OUT/dp_tests.log | head -30
