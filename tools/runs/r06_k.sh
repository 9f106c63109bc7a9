OUT=gpurun_out/r06_k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_extreme_scales.py -q -m gpu -k "canvas or st_write or extreme or grid_stride" > $OUT/canvas_tests.log 2>&1; tail -3 $OUT/canvas_tests.log; grep -E "^(FAILED|ERROR)" $OUT/canvas_tests.log | head -40
timeout 900 python tools/probes/canvas_gs_ab.py > $OUT/canvas_gs_ab.txt 2>&1; grep -v "   64 " $OUT/canvas_gs_ab.txt | head -20
