OUT=gpurun_out/r06_i; mkdir -p $OUT
T=tools/kbench/bin/st_trace
( echo "== c4 b2048 narrow"; $T 2048 5 100 28 1 4 0.45 0.65 ) > $OUT/st_trace_im.txt 2>&1
grep -B1 -A13 "trace canvas_unroll_bwd  " $OUT/st_trace_im.txt
