#!/bin/bash
# per-position trace of the replayed step at configs[1] on both submission paths (what does the runtime setting change, node by node?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03_pos; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  rm -rf $O/trace_$V
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=$V rocprofv3 --kernel-trace -d $O/trace_$V -o b -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 600 --warmup 50 > /dev/null 2> $O/trace_$V.log
  python $ROOT/tools/rocpd_summary.py $(find $O/trace_$V -name "*.db" | head -1) --by-position step_epilogue_kernel --every 1 > $O/positions_c2_b64_packet_capture_$V.txt
  rm -rf $O/trace_$V
done
head -3 $O/positions_c2_b64_packet_capture_0.txt; head -2 $O/positions_c2_b64_packet_capture_1.txt
