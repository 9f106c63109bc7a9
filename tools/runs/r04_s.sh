#!/bin/bash
# which instruction classes the canvas kernels spend their issue slots on, throughput regime (8192 images x 3 glimpses)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04_s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > $O/counters.txt
wc -l $O/counters.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o r -- python $ROOT/tools/probes/canvas_pmc.py > $O/p$i.log 2>&1 || echo "pass $i ($C) failed"
done
python $ROOT/tools/rocpd_pmc.py $(find $O -name "*.db" | sort) | grep -v "^void at::\|elementwise\|distribution" > $O/canvas_pmc.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $ROOT/tools/probes/canvas_pmc.py > $O/t.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find $O/t -name "*.db" | head -1) > $O/canvas_stats.txt
rm -rf $O/p*/ $O/t
cat $O/canvas_pmc.txt | grep "st_write\|st_read\|kernel " ; grep "st_write\|st_read" $O/canvas_stats.txt
