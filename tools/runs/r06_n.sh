OUT=gpurun_out/r06_n; mkdir -p $OUT
for rep in 1 2; do for G in 0 512 1024 1536 2048 16384 32768 65536; do
  AIR_ST_READ_GRID=$G timeout 120 python tools/probes/read_c4_grid.py 2>/dev/null | tail -1
done; done | tee $OUT/read_c4_grid2.txt
