#!/bin/bash
# rocprofv3 kernel trace of scripts/latency_bench.py: the launches of the order-parallel form (set-up, order tasks, replay) per solve
OUT=$PWD/gpurun_out/prof/lat
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o lat -- python3 $REPO/scripts/latency_bench.py > $OUT/lat.log 2>&1
cd $REPO
grep bins $OUT/lat.log
python3 - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
q = "select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%k_sos_stream%' or name like '%k_profile%' or name like '%aggregate%' group by name order by sum(duration) desc"
for r in con.execute(q):
    print("%-72s calls %4d  avg %9.1f us  min %9.1f  max %9.1f" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
PY
