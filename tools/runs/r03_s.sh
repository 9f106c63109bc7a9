#!/bin/bash
O=gpurun_out/r03_s; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py tests/test_bench_multirank.py -m gpu -x -q -k "data_parallel or bench" > $O/dp.log 2>&1; tail -6 $O/dp.log
python -c "from attend_infer_repeat_amd import build; print(build.source_digest()[:12])"
python bench.py --no-cpu-baseline --no-sweep > $O/bench_c2.json 2> $O/bench.log; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['config']['collective'])"
