# Round 6: full GPU suite, then the profiler passes + unprofiled lines of this binary (tag r06_r)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_r; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log | head -2; grep -E "^(FAILED|ERROR)" $OUT/gpu_tests.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 bash tools/profile_round.sh r06_r pmc > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
for n in c2_b64 c4_b64 c5_b1024; do echo "== $n"; head -3 $OUT/r06_r_positions_$n.txt; sort -k3 -n -r $OUT/r06_r_positions_$n.txt | head -8; done
