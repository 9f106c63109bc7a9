#!/bin/bash
O=gpurun_out/r03_o; mkdir -p $O
timeout 1200 python -m pytest tests/test_bench_multirank.py -m gpu -x -q > $O/multirank.log 2>&1; tail -15 $O/multirank.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --no-cpu-baseline --no-sweep > $O/bench_c2.json 2> $O/bench.log; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['roofline']['traffic'])"
