# rider sizing (AIR_RIDER_QPT float4 per rider thread): same-box A/B at configs[1] / configs[3] + positions at the best setting
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_p; mkdir -p $OUT
for i in 1 2; do
  for V in 2 1 0.5 0.25; do
    AIR_RIDER_QPT=$V timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-sweep --steps 3000 --warmup 200 2>/dev/null | tail -1 > $OUT/c2_q${V}_$i.json
    AIR_RIDER_QPT=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_q${V}_$i.json
    python - <<PY
import json
for c in ("c2", "c4"):
    d=json.load(open("$OUT/%s_q${V}_$i.json" % c)); print(c, "qpt=$V run $i", d["ms_per_step"], d["value"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for V in 1 0.5; do
AIR_RIDER_QPT=$V timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o b -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --no-other-configs --steps 1500 --warmup 100 > $OUT/profiled.json 2> $OUT/profiled.log
python $ROOT/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-position gemm_grouped_opt_kernel --every 2 > $OUT/positions_c2_q$V.txt 2>&1
rm -rf $OUT/trace
cut -c1-100 $OUT/positions_c2_q$V.txt | sed -n 24,32p
done
