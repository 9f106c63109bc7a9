B="python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-sweep --no-other-configs"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
$B 2>/dev/null | show feeder4096
$B --dataset-images 64 2>/dev/null | show feeder64
$B --dataset-images 256 2>/dev/null | show feeder256
$B --dataset-images 65536 2>/dev/null | show feeder65536
$B --fixed-batch 2>/dev/null | show fixed
done
