#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box:  bash tools/profile_round.sh <tag>
#   1. the unprofiled bench line (full: sweeps + CPU baseline)             -> <tag>_bench_c2_b64_unprofiled.json
#   2. kernel trace of the default bench (B=64, hipGraph replay)           -> <tag>_bench_c2_b64_kernel_stats.txt
#   3. PMC passes (own runs, --kernel-trace only, as gpurun requires)      -> <tag>_bench_c2_b64_pmc.txt, <tag>_instep_pmc.json
#   4. bench lines of the other named configurations                       -> <tag>_bench_c4_b64.json, <tag>_bench_c5_b1024_bf16.json, ...
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
set -u
TAG=${1:-r02_x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY="python $ROOT/bench.py"
$PY > "$OUT/${TAG}_bench_c2_b64_unprofiled.json" 2> "$OUT/bench.log"
# bench lines of the other named configurations first: the profiler passes below leave the box in a state in which the next
# unprofiled run can come out 10 % slow (seen once on the batch-1024 line)
$PY --config c4 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c4_b64.json" 2>> "$OUT/bench.log"
$PY --config c5 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c5_b1024_bf16.json" 2>> "$OUT/bench.log"
$PY --batch 1024 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c2_b1024_f32.json" 2>> "$OUT/bench.log"
$PY --batch 512 --no-cpu-baseline --no-sweep --steps 500 --warmup 50 > "$OUT/${TAG}_bench_c2_b512_f32.json" 2>> "$OUT/bench.log"
$PY --batch 256 --no-cpu-baseline --no-sweep --steps 500 --warmup 50 > "$OUT/${TAG}_bench_c2_b256_f32.json" 2>> "$OUT/bench.log"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $PY --no-cpu-baseline --no-sweep > "$OUT/${TAG}_bench_c2_b64_profiled.json" 2> "$OUT/trace.log"
python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace" -name "*.db" | head -1) --steps 2600 > "$OUT/${TAG}_bench_c2_b64_kernel_stats.txt"
python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace" -name "*.db" | head -1) --by-position step_epilogue_kernel > "$OUT/${TAG}_bench_c2_b64_positions.txt"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_$N" -o r -- $PY --no-cpu-baseline --no-sweep --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_$N.log"
done
python $ROOT/tools/rocpd_pmc.py $(find "$OUT" -path "*pmc_*" -name "*.db" | sort) > "$OUT/${TAG}_bench_c2_b64_pmc.txt"
DIGEST=$(cd $ROOT && python -c "from attend_infer_repeat_amd import build; print(build.source_digest())")
python $ROOT/tools/pmc_to_json.py --fetch $(find "$OUT/pmc_FETCH_SIZE" -name "*.db" | head -1) --write $(find "$OUT/pmc_WRITE_SIZE" -name "*.db" | head -1) \
    --digest $DIGEST --shape 50 50 20 20 3 64 > "$OUT/${TAG}_instep_pmc.json"
# the throughput regime: per-position picture of the replayed step at batch 1024 (bf16 operands = configs[4], and fp32)
for V in "c5:--config c5" "b1024_f32:--batch 1024"; do
  N=${V%%:*}; A=${V#*:}
  rm -rf "$OUT/trace_$N"
  rocprofv3 --kernel-trace -d "$OUT/trace_$N" -o b -- $PY $A --no-cpu-baseline --no-sweep --steps 200 --warmup 20 > /dev/null 2>> "$OUT/trace.log"
  python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace_$N" -name "*.db" | head -1) --by-position step_epilogue_kernel > "$OUT/${TAG}_positions_$N.txt"
  python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace_$N" -name "*.db" | head -1) > "$OUT/${TAG}_bench_${N}_kernel_stats.txt"
  rm -rf "$OUT/trace_$N"
done
rm -rf "$OUT"/trace "$OUT"/pmc_*/   # the SQLite traces are large; the text summaries are what travels back
ls -la "$OUT"
