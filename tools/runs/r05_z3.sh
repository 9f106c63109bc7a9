# Round 5, last session: per-position kernel durations of configs[3] for two engine seeds x streaming short-K on / off (second half of r05_z2.sh)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_z3; mkdir -p $OUT
P="timeout 120 python $ROOT/tools/probes/placement_probe.py"
cd /tmp && export TMPDIR=/tmp
for S in 1 1000004; do for V in 1 0; do
  rm -rf $OUT/trace
  AIR_PROBE_SEED=$S AIR_GEMM_SHORTK=$V timeout 200 rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $ROOT/tools/probes/placement_probe.py c4 fresh 300 > /dev/null 2> $OUT/prof.log
  python $ROOT/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-position gemm_grouped_opt --every 2 > $OUT/positions_seed${S}_shortk$V.txt 2>&1
  rm -rf $OUT/trace
done; done
python - <<PY
import re
def load(p):
    rows = []
    for l in open(p):
        m = re.match(r"\s*(\d+) (\S.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: rows.append((m.group(2).strip(), float(m.group(3)), float(m.group(5))))
    return rows
for V in (1, 0):
    a, b = load("$OUT/positions_seed1_shortk%d.txt" % V), load("$OUT/positions_seed1000004_shortk%d.txt" % V)
    print("shortk=%d: positions %d / %d; sum of kernel us seed 1: %.1f  seed 1000004: %.1f; sum of gaps: %.1f / %.1f" % (V, len(a), len(b), sum(x[1] for x in a), sum(x[1] for x in b), sum(x[2] for x in a), sum(x[2] for x in b)))
    for i, (x, y) in enumerate(zip(a, b)):
        if abs(x[1] - y[1]) > 0.8 or abs(x[2] - y[2]) > 0.8:
            print("  pos %2d %-44s seed1 %6.2f us (gap %5.2f)   seed1000004 %6.2f us (gap %5.2f)" % (i, x[0][:44], x[1], x[2], y[1], y[2]))
PY
