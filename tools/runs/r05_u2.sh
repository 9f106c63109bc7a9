# Round 5, final tree: the whole GPU suite once more (after the rendezvous-port fix of the spawning tests)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_u2; mkdir -p $OUT
timeout 660 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/gpu_tests.log | head
