#!/bin/bash
# HIP runtime knobs that could change what a captured kernel chain costs per node / per replay (configs[1], default bench); every run
# under its own timeout (ROC_SYSTEM_SCOPE_SIGNAL=0 hung the first sweep)
O=gpurun_out/r03_env; mkdir -p $O
run() { timeout 90 env "$@" python bench.py --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])
except Exception as e: print('$*', 'FAILED', e)"; }
{
run A=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_GRAPH_BATCH_SIZE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run AMD_DIRECT_DISPATCH=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_MAX_BATCH_SIZE=4096
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 ROC_USE_FGS_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=0
run A=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
} | tee $O/env_sweep2.txt
for C in c4 c5; do for V in 1 0; do timeout 120 env DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V python bench.py --config $C --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C packet_capture=$V', d['value'], d['ms_per_step'])"; done; done | tee -a $O/env_sweep2.txt
