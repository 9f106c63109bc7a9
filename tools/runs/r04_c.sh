#!/bin/bash
# round 4, call c: the early engine-only non-finite update of seed 9 on the round-3 dataset, with per-buffer diagnostics
O=gpurun_out/r04_c; mkdir -p $O
timeout 900 python tools/blowup_replay.py --seed 9 --legacy-data --out $O/blowup > $O/blowup_seed9_legacy.log 2>&1; echo "seed 9 rc=$?"; tail -14 $O/blowup_seed9_legacy.log
rm -f $O/blowup/*.pt
