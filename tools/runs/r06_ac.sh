timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attend_fwd_image_major or st_read" 2>&1 | tail -3
for L in 1 0; do
echo "== lean=$L B=1024"
AIR_ATTEND_LEAN=$L ONLY=attend PREC=1 timeout 60 tools/kbench/bin/st_trace 1024 3 50 20 | grep -v amdgpu.ids
AIR_ATTEND_LEAN=$L ONLY=attend PREC=1 timeout 60 tools/kbench/bin/st_trace_tr 1024 3 50 20 | grep -v amdgpu.ids
done
echo "== c2"; ONLY=attend timeout 60 tools/kbench/bin/st_trace_tr 64 3 50 20 | grep -v amdgpu.ids
echo "== c4"; ONLY=attend timeout 60 tools/kbench/bin/st_trace_tr 64 5 100 28 | grep -v amdgpu.ids
