#!/bin/bash
# instruction-class counters of the ST read after the vectorised rewrite (same probe as r04_s)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04_u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  B=${B:-8192} timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o r -- python $ROOT/tools/probes/canvas_pmc.py > $O/p$i.log 2>&1 || echo "pass $i ($C) failed"
done
python $ROOT/tools/rocpd_pmc.py $(find $O -name "*.db" | sort) | grep "st_read\|^kernel" > $O/read_pmc.txt
B=${B:-8192} timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $ROOT/tools/probes/canvas_pmc.py > $O/t.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find $O/t -name "*.db" | head -1) | grep "st_read" >> $O/read_pmc.txt
rm -rf $O/p*/ $O/t
cat $O/read_pmc.txt
