#!/bin/bash
# plan / dispatch switches at the configurations they were NOT tuned on (defaults were measured at configs[1] and configs[4])
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04_z
run() {  # run <config args> -- ENV=..
  local CFG="$1"; shift
  local ms=$(env "$@" python bench.py $CFG --no-cpu-baseline --no-sweep --steps 600 --warmup 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('kernel_launches_per_step'))")
  echo "$CFG | $* | $ms"
}
{
for V in X=0 AIR_SPLIT_K0=0 AIR_OPT_RIDERS=0 AIR_FUSE_ATTEND_M=0 AIR_FUSE_LSTM_TILES=0 AIR_CANVAS_SPLIT=1 X=0; do run "--config c4" $V; done
for V in X=0 AIR_FUSE_ATTEND_M=0 AIR_FUSE_LSTM_WIDE=0 AIR_GEMM_BIG_XCD=0 AIR_GEMM_WIDE_NT_K=256 AIR_GEMM_WIDE_NT_K=1024 AIR_GEMM_WIDE_TN_BF16=24 AIR_GEMM_WIDE_TN_BF16=96 AIR_GEMM_WIDE_MIN_TILES=500 AIR_GEMM_WIDE_MIN_TILES=2000 AIR_OPT_RIDERS=0 AIR_FUSE_CANVAS_THROUGHPUT=1 X=0; do run "--config c5" $V; done
for V in X=0 AIR_FUSE_ATTEND_M=0 AIR_OPT_RIDERS=0 AIR_DEFER_DW_MIN_ROWS=100000 X=0; do run "--batch 256" $V; done
} 2>&1 | tee gpurun_out/r04_z/switch_sweep.txt
