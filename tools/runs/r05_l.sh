# positions of the replayed step at configs[1] on the current tree (kernel trace only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --no-other-configs --steps 1500 --warmup 100 > $OUT/profiled.json 2> $OUT/profiled.log
DB=$(find $OUT/trace -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB --by-position gemm_grouped_opt_kernel --every 2 > $OUT/positions_c2.txt 2>&1
rm -rf $OUT/trace
cut -c1-110 $OUT/positions_c2.txt | head -40
