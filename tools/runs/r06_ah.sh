timeout 900 python -m pytest tests/test_engine.py -x -q -m gpu -k "many_ranks" --durations=12 > gpurun_out/r06_ah.log 2>&1; grep -n "Error\|error\|Traceback\|assert" gpurun_out/r06_ah.log | head -30
