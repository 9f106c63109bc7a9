# Round 5, last session: buffer placement probe (tools/probes/placement_probe.py) -- contexts x flat layouts, configs[3] and configs[1]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_z; mkdir -p $OUT
P="timeout 120 python tools/probes/placement_probe.py"
{
for i in 1 2; do
  for CTX in fresh cached second; do
    $P c4 $CTX 2>/dev/null | tail -1
    AIR_GEMM_SHORTK=0 $P c4 $CTX 2>/dev/null | tail -1
  done
  for L in packed stagger:0 stagger:4096 stagger:65536 stagger:8448; do
    AIR_FLAT_LAYOUT=$L $P c4 fresh 2>/dev/null | tail -1
    AIR_FLAT_LAYOUT=$L $P c4 second 2>/dev/null | tail -1
  done
  for CTX in fresh cached; do $P c2 $CTX 2000 2>/dev/null | tail -1; done
  for L in packed stagger:0 stagger:4096 stagger:65536 stagger:8448; do
    AIR_FLAT_LAYOUT=$L $P c2 fresh 2000 2>/dev/null | tail -1
  done
done
} | tee $OUT/placement.txt
