# positions of the replayed step at configs[1] with / without the BPTT-entry fold (kernel trace only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  AIR_LSTM_BWD_ENTRY=$V timeout 600 rocprofv3 --kernel-trace -d $OUT/trace_$V -o b -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --no-other-configs --steps 1500 --warmup 100 > $OUT/profiled_$V.json 2> $OUT/profiled_$V.log
  DB=$(find $OUT/trace_$V -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB --by-position gemm_grouped_opt_kernel --every 2 > $OUT/positions_c2_entry$V.txt 2>&1
  rm -rf $OUT/trace_$V
  cut -c1-100 $OUT/positions_c2_entry$V.txt | sed -n 22,34p
done
