R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config c4 --steps 600 --warmup 60 --no-cpu-baseline --no-sweep --no-other-configs"
for v in 512 1024 512 1024; do
  if [ $v = 1024 ]; then export AIR_ATTEND_FWD_1024=1 AIR_ATTEND_BWD_1024=1; else unset AIR_ATTEND_FWD_1024 AIR_ATTEND_BWD_1024; fi
  rm -rf /tmp/prof_$v
  timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_$v -o p -- $B > /tmp/b_$v.log 2>&1 < /dev/null
  echo "== $v"
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  if [ -n "$DB" ]; then timeout 100 python $R/tools/rocpd_summary.py $DB < /dev/null | grep -E "attend|st_write|TOTAL|total" | head; fi
done
