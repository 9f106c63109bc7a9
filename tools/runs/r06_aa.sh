for P in 0 1; do
echo "== PREC=$P B=1024"; ONLY=st_read_fwd PREC=$P tools/kbench/bin/st_trace 1024 3 50 20 | grep -v amdgpu.ids
ONLY=attend PREC=$P tools/kbench/bin/st_trace 1024 3 50 20 | grep -v amdgpu.ids
ONLY=attend PREC=$P tools/kbench/bin/st_trace_tr 1024 3 50 20 | grep -v amdgpu.ids
done
echo "== c4 B=64 T=5"; ONLY=attend tools/kbench/bin/st_trace_tr 64 5 100 28 | grep -v amdgpu.ids
echo "== c2 B=64 T=3"; ONLY=attend tools/kbench/bin/st_trace_tr 64 3 50 20 | grep -v amdgpu.ids
