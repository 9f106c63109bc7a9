#!/bin/bash
# second sweep of runtime knobs (queues, interrupts, waits), on top of the setting the package applies; each run under its own timeout
O=gpurun_out/r03_env; mkdir -p $O
run() { timeout 90 env "$@" python bench.py --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])
except Exception as e: print('$*', 'FAILED', e)"; }
{
run A=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1
run HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0
run A=1
} | tee $O/env_sweep3.txt
