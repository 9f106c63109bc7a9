#!/bin/bash
# does the ordinary submission path (graph packet capture off) change what graph branches cost?  probe + the two-lane step
O=gpurun_out/r03_rt2; mkdir -p $O
for V in 1 0; do echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V timeout 120 tools/kbench/bin/graph_branch 300 64; done > $O/graph_branch.txt 2>&1
for V in 1 0; do echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V timeout 120 tools/kbench/bin/graph_nodes; done > $O/graph_nodes.txt 2>&1
run() { timeout 120 env "$@" python bench.py --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['config']['kernel_launches_by_lane'])
except Exception as e: print('$*', 'FAILED', e)"; }
{
run A=0
run AIR_TWO_LANE=1
run AMD_DIRECT_DISPATCH=0
run AIR_RUNTIME_ENV=0 AIR_TWO_LANE=1
run A=1
} | tee $O/two_lane.txt
