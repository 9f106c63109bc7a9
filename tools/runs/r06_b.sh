# Round 6: phase stamps of the glimpse-space canvas backward (tools/kbench/st_trace, -DAIR_TRACE)
OUT=gpurun_out/r06_b; mkdir -p $OUT
T=tools/kbench/bin/st_trace
( echo "== c4 b64 narrow"; $T 64 5 100 28 1 4 0.45 0.65; echo "== c4 b64 wide"; $T 64 5 100 28 1 4 1.4 2.8; echo "== c2 b64 narrow"; $T 64 3 50 20 1 4 0.45 0.65;
  echo "== c2 b64 wide"; $T 64 3 50 20 1 4 1.4 2.8 ) > $OUT/st_trace.txt 2>&1
grep -v "st_read\|^$" $OUT/st_trace.txt | head -150
