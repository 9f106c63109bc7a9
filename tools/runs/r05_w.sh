# streaming short-K weight gradients: workgroups per problem (AIR_GEMM_SHORTK_WGS) at configs[3], same box
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_w; mkdir -p $OUT
for i in 1 2; do
  for V in 256 128 384 512 640; do
    AIR_GEMM_SHORTK_WGS=$V timeout 300 python bench.py --config c4 --no-other-configs --no-cpu-baseline --no-sweep --steps 2000 --warmup 200 2>/dev/null | tail -1 > $OUT/c4_w${V}_$i.json
    python -c "import json; d=json.load(open('$OUT/c4_w${V}_$i.json')); print('c4 wgs=$V run $i', d['ms_per_step'], d['value'])"
  done
done
