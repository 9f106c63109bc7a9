#!/bin/bash
# where do the deferred weight-gradient launches of the batch-1024 bf16 step wait?  L2 hit rate, fabric read requests, wave stall cycles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03_w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC|TCP|SQ|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/counters_avail.txt
PY="python $ROOT/bench.py --config c5 --no-cpu-baseline --no-sweep --steps 20 --warmup 5"
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o r -- $PY > /dev/null 2> $O/p$i.log || echo "pass $i ($C) failed" >> $O/fail.txt
done
python $ROOT/tools/rocpd_pmc.py $(find $O -name "*.db" | sort) | grep -E "^kernel|^#|big_group|lstm_bwd_wide16|gemm_grouped_wide16_kernel<1, 8, false, false, true, true>" > $O/dw_pmc.txt
rm -rf $O/p*/
cat $O/dw_pmc.txt | head -80; cat $O/fail.txt 2>/dev/null
