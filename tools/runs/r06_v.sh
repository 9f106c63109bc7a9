OUT=gpurun_out/r06_v; mkdir -p $OUT
run() { env "$@" timeout 300 python bench.py $CFG --fixed-batch --no-cpu-baseline --no-sweep --no-other-configs --steps 2000 --warmup 200 2>/dev/null > $OUT/line.json; python -c "import json; d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('$CFG $*', d['ms_per_step'], d['config']['kernel_launches_per_step'])"; }
for rep in 1 2; do
CFG="" ; run AIR_X=0; run AIR_CANVAS_SPLIT=2
CFG="--config c4"; run AIR_X=0; run AIR_FUSE_CANVAS=1
done
