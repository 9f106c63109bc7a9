timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
B="--steps 2000 --warmup 200 --no-cpu-baseline --no-sweep --no-other-configs"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
timeout 300 python bench.py $B 2>/dev/null | show c2
timeout 300 python bench.py --config c4 $B 2>/dev/null | show c4
AIR_ATTEND_FWD_1024=1 timeout 300 python bench.py --config c4 $B 2>/dev/null | show c4_fwd1024
timeout 300 python bench.py --config c5 $B 2>/dev/null | show c5
AIR_ATTEND_LEAN=0 timeout 300 python bench.py --config c5 $B 2>/dev/null | show c5_nolean
done
