#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box:  bash tools/profile_round.sh <tag> [pmc|bench|all]
#   pmc   : PMC passes (own runs, --kernel-trace only, as gpurun requires) of the default bench at every named shape
#           -> <tag>_<cfg>_instep_pmc.json (digest- and shape-matched: bench.py reads `roofline.traffic` from them once they are
#              copied into profiles/), <tag>_bench_c2_b64_pmc.txt; kernel trace of the default bench -> *_kernel_stats.txt,
#              per-position traces of the replayed step at every named shape -> <tag>_positions_<cfg>.txt
#   bench : the unprofiled bench lines of every named configuration (run AFTER the pmc JSONs of the same build are in profiles/,
#           so that no committed line carries traffic = null)
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/.
set -u
TAG=${1:-r03_x}
MODE=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY="python $ROOT/bench.py"
DIGEST=$(cd $ROOT && python -c "from attend_infer_repeat_amd import build; print(build.source_digest())")

pmc_pass() {      # pmc_pass <cfg name> "<H W h w T B>" <bench args...>
  local NAME=$1 SHAPE=$2; shift 2
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/pmc_${NAME}_$C"
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_${NAME}_$C" -o r -- $PY "$@" --no-cpu-baseline --no-sweep --no-other-configs --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_${NAME}_$C.log"
  done
  python $ROOT/tools/pmc_to_json.py --fetch $(find "$OUT/pmc_${NAME}_FETCH_SIZE" -name "*.db" | head -1) \
      --write $(find "$OUT/pmc_${NAME}_WRITE_SIZE" -name "*.db" | head -1) --digest $DIGEST --shape $SHAPE > "$OUT/${TAG}_${NAME}_instep_pmc.json"
}
positions() {     # positions <cfg name> "<H W h w T B>" <bench args...>
  local NAME=$1 SHAPE=$2; shift 2
  rm -rf "$OUT/trace_$NAME"
  rocprofv3 --kernel-trace -d "$OUT/trace_$NAME" -o b -- $PY "$@" --no-cpu-baseline --no-sweep --no-other-configs --steps 300 --warmup 30 > /dev/null 2>> "$OUT/trace.log"
  local DB=$(find "$OUT/trace_$NAME" -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $DB --by-position ${ANCHOR:-step_epilogue_kernel} --every ${EVERY_C4:-1} --json "$OUT/${TAG}_${NAME}_instep_durations.json" --digest $DIGEST --dtype ${DTYPE:-f32} --shape $SHAPE > "$OUT/${TAG}_positions_$NAME.txt"
  python $ROOT/tools/rocpd_summary.py $DB > "$OUT/${TAG}_bench_${NAME}_kernel_stats.txt"
  rm -rf "$OUT/trace_$NAME"
}

if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
  pmc_pass c2_b64 "50 50 20 20 3 64"
  pmc_pass c4_b64 "100 100 28 28 5 64" --config c4
  pmc_pass c2_b1024 "50 50 20 20 3 1024" --batch 1024
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --pmc $C --kernel-trace -d "$OUT/pmc_c2_b64_$N" -o r -- $PY --no-cpu-baseline --no-sweep --no-other-configs --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_$N.log"
  done
  python $ROOT/tools/rocpd_pmc.py $(find "$OUT" -path "*pmc_c2_b64_*" -name "*.db" | sort) > "$OUT/${TAG}_bench_c2_b64_pmc.txt"
  python $ROOT/tools/rocpd_pmc.py --mfma-json $(find "$OUT/pmc_c2_b64_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES" -name "*.db" | head -1) $DIGEST f32 50 50 20 20 3 64 > "$OUT/${TAG}_c2_b64_mfma_util.json"
  for N in c5_b1024; do :; done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc_c5_mfma" -o r -- $PY --config c5 --no-cpu-baseline --no-sweep --no-other-configs --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_c5_mfma.log"
  python $ROOT/tools/rocpd_pmc.py $(find "$OUT/pmc_c5_mfma" -name "*.db" | sort) > "$OUT/${TAG}_bench_c5_b1024_pmc.txt"
  python $ROOT/tools/rocpd_pmc.py --mfma-json $(find "$OUT/pmc_c5_mfma" -name "*.db" | head -1) $DIGEST bf16 50 50 20 20 3 1024 > "$OUT/${TAG}_c5_b1024_mfma_util.json"
  # kernel trace + per-position picture of the replayed step at every named shape
  rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $PY --no-cpu-baseline --no-sweep --no-other-configs > "$OUT/${TAG}_bench_c2_b64_profiled.json" 2> "$OUT/trace.log"
  python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace" -name "*.db" | head -1) --steps 2600 > "$OUT/${TAG}_bench_c2_b64_kernel_stats.txt"
  # (round 5: the latency-regime step ends with the weight-gradient launch that carries the folded update)
  python $ROOT/tools/rocpd_summary.py $(find "$OUT/trace" -name "*.db" | head -1) --by-position gemm_grouped_opt_kernel --every ${EVERY_C2:-2} --json "$OUT/${TAG}_c2_b64_instep_durations.json" --digest $DIGEST --shape 50 50 20 20 3 64 > "$OUT/${TAG}_positions_c2_b64.txt"
  ANCHOR=gemm_grouped_opt_kernel EVERY_C4=${EVERY:-1} positions c4_b64 "100 100 28 28 5 64" --config c4
  DTYPE=bf16 positions c5_b1024 "50 50 20 20 3 1024" --config c5
  positions c2_b1024_f32 "50 50 20 20 3 1024" --batch 1024
  rm -rf "$OUT"/trace "$OUT"/pmc_*/   # the SQLite traces are large; the text summaries are what travels back
fi
if [ "$MODE" = bench ] || [ "$MODE" = all ]; then
  $PY > "$OUT/${TAG}_bench_c2_b64_unprofiled.json" 2> "$OUT/bench.log"
  $PY --steps 20 --warmup 5 > "$OUT/${TAG}_bench_c2_b64_driver_command.json" 2>> "$OUT/bench.log"
  $PY --config c4 --no-cpu-baseline --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c4_b64.json" 2>> "$OUT/bench.log"
  $PY --config c5 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c5_b1024_bf16.json" 2>> "$OUT/bench.log"
  $PY --batch 1024 --no-cpu-baseline --no-sweep --steps 1000 --warmup 100 > "$OUT/${TAG}_bench_c2_b1024_f32.json" 2>> "$OUT/bench.log"
  $PY --batch 512 --no-cpu-baseline --no-sweep --steps 500 --warmup 50 > "$OUT/${TAG}_bench_c2_b512_f32.json" 2>> "$OUT/bench.log"
  $PY --batch 256 --no-cpu-baseline --no-sweep --steps 500 --warmup 50 > "$OUT/${TAG}_bench_c2_b256_f32.json" 2>> "$OUT/bench.log"
fi
ls -la "$OUT"
