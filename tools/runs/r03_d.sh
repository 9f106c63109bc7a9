#!/bin/bash
# round-3 GPU session D: full GPU suite, f1 training evidence, per-position traces of the bf16 data path
O=gpurun_out/r03_d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
bash tools/runs/r03_train.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "c5_storage:--config c5" "c5_round_only:--config c5"; do
  N=${V%%:*}; A=${V#*:}
  [ $N = c5_round_only ] && export AIR_BF16_STORAGE=0
  rm -rf $R/$O/trace_$N
  rocprofv3 --kernel-trace -d $R/$O/trace_$N -o b -- python $R/bench.py $A --no-cpu-baseline --no-sweep --steps 200 --warmup 20 > /dev/null 2>> $R/$O/trace.log
  python $R/tools/rocpd_summary.py $(find $R/$O/trace_$N -name "*.db" | head -1) --by-position step_epilogue_kernel > $R/$O/positions_$N.txt
  rm -rf $R/$O/trace_$N
  unset AIR_BF16_STORAGE
done
head -50 $R/$O/positions_c5_storage.txt
