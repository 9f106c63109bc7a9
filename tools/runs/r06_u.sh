OUT=gpurun_out/r06_u; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_engine.py -q -m gpu -s -k "many_ranks or two_ranks or ipc_barrier" > $OUT/dp_tests.log 2>&1; grep -E "passed|failed|flag block|^FAILED|^ERROR" $OUT/dp_tests.log | tail -12
AIR_IPC_FINEGRAINED=0 timeout 600 python -m pytest tests/test_engine.py -q -m gpu -s -k "many_ranks and ipc and 4" 2>&1 | grep -E "passed|failed|flag block" | tail -3
