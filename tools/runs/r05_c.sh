ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_c; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for M in base warm; do
  rocprofv3 --kernel-trace -d $OUT/trace_$M -o b -- python $ROOT/tools/probes/cold_start_probe.py $M 400 > $OUT/probe_$M.json 2> $OUT/probe_$M.log
  DB=$(find $OUT/trace_$M -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB --by-position step_epilogue_kernel > $OUT/positions_$M.txt
  rm -rf $OUT/trace_$M
done
python $ROOT/tools/probes/cold_start_probe.py base 2000 > $OUT/unprofiled_base.json 2>/dev/null
python $ROOT/tools/probes/cold_start_probe.py warm 2000 > $OUT/unprofiled_warm.json 2>/dev/null
head -c 300 $OUT/probe_warm.json; echo; tail -3 $OUT/probe_warm.log; head -3 $OUT/positions_warm.txt; cat $OUT/unprofiled_*.json | cut -c1-80
