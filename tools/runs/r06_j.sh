OUT=gpurun_out/r06_j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_extreme_scales.py -q -m gpu -k "canvas or st_write or extreme or grid_stride" > $OUT/canvas_tests.log 2>&1; tail -3 $OUT/canvas_tests.log; grep -E "^(FAILED|ERROR)" $OUT/canvas_tests.log | head -40
timeout 900 python tools/probes/canvas_gs_ab.py > $OUT/canvas_gs_ab.txt 2>&1; cat $OUT/canvas_gs_ab.txt
T=tools/kbench/bin/st_trace
( echo "== c4 b2048 narrow"; $T 2048 5 100 28 1 4 0.45 0.65; echo "== c2 b4096 narrow"; $T 4096 3 50 20 1 4 0.45 0.65;  echo "== c2 b64 narrow"; $T 64 3 50 20 1 4 0.45 0.65 ) > $OUT/st_trace_im.txt 2>&1
grep -B1 -A10 "trace canvas_unroll_bwd  \|trace canvas_fused" $OUT/st_trace_im.txt | grep -v "phase 9"
