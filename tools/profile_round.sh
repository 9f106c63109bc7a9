#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box:  bash tools/profile_round.sh <tag>
#   1. kernel trace of the default bench (B=64, hipGraph replay)        -> <tag>_bench_c2_b64_kernel_stats.txt
#   2. PMC passes (own runs, --kernel-trace only, as gpurun requires)   -> <tag>_bench_c2_b64_pmc.txt
#   3. the unprofiled bench line                                         -> <tag>_bench_c2_b64_unprofiled.json
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
set -u
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY="python $ROOT/bench.py"
$PY --no-cpu-baseline > "$OUT/${TAG}_bench_c2_b64_unprofiled_nocpu.json" 2> "$OUT/bench_nocpu.log"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $PY --no-cpu-baseline --no-sweep > "$OUT/${TAG}_bench_c2_b64_profiled.json" 2> "$OUT/trace.log"
python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace" -name "*.db" | head -1) --steps 2200 > "$OUT/${TAG}_bench_c2_b64_kernel_stats.txt"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$N" -o r -- $PY --no-cpu-baseline --no-sweep --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_$N.log"
done
python $ROOT/tools/rocpd_pmc.py $(find "$OUT" -path "*pmc_*" -name "*.db" | sort) > "$OUT/${TAG}_bench_c2_b64_pmc.txt"
rm -rf "$OUT"/trace "$OUT"/pmc_*/   # the SQLite traces are large; the text summaries are what travels back
ls -la "$OUT"
