ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_o; mkdir -p $OUT
PYTHONPATH=$ROOT python tools/probes/plan_dump.py c5 > $OUT/plan_c5.txt 2>&1; cat $OUT/plan_c5.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_c5
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace_c5 -o b -- python $ROOT/bench.py --config c5 --fixed-batch --no-cpu-baseline --no-sweep --no-other-configs --steps 300 --warmup 30 > /dev/null 2> $OUT/trace.log
DB=$(find $OUT/trace_c5 -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py $DB --by-position step_epilogue_kernel > $OUT/positions_c5.txt; head -45 $OUT/positions_c5.txt
python - <<PY
import sqlite3
con = sqlite3.connect("$DB")
for t in ("memory_copies", "memory_copy", "rocpd_memory_copy"):
    try:
        print(t, con.execute("select count(*) from %s" % t).fetchall())
    except Exception as e:
        print(t, "-", e)
print([r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")][:60])
PY
rm -rf $OUT/trace_c5
cd $ROOT
timeout 300 python tools/probes/read_c4_grid.py 2>/dev/null | tail -1
