# Round 6, first session: canvas tests on the glimpse-space backward + the old/new A/B sweep.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_extreme_scales.py -q -m gpu -k "canvas or st_write or extreme or grid_stride" > $OUT/canvas_tests.log 2>&1; tail -5 $OUT/canvas_tests.log; grep -E "^(FAILED|ERROR)" $OUT/canvas_tests.log | head -40
timeout 900 python tools/probes/canvas_gs_ab.py gsim:AIR_CANVAS_IMG_MIN_UNITS=0 gsim512:AIR_CANVAS_IMG_MIN_UNITS=0,AIR_CANVAS_GS_THREADS=512 gsum:AIR_CANVAS_BWD_IMG=0 > $OUT/canvas_gs_ab.txt 2>&1; cat $OUT/canvas_gs_ab.txt
