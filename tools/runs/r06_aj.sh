echo "== c4 b2048 narrow (final binary of round 6)"; ONLY="canvas_unroll_bwd" timeout 120 tools/kbench/bin/st_trace_tr 2048 5 100 28 | grep -v amdgpu.ids
echo "== c4 b64 narrow (final binary of round 6)"; ONLY="canvas" timeout 120 tools/kbench/bin/st_trace_tr 64 5 100 28 | grep -v amdgpu.ids
echo "== c2 b64 narrow (final binary of round 6)"; ONLY="canvas" timeout 120 tools/kbench/bin/st_trace_tr 64 3 50 20 | grep -v amdgpu.ids
