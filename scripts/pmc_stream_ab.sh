#!/bin/bash
# Fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the streamed solver in its two launch forms, realistic mix.
OUT=$PWD/gpurun_out/prof/stream_ab
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
ARGR="$REPO/bench.py --steps 2 --warmup 1 --no-cpu --workload realistic --bins 4096"
for p in 1 0; do
  export SOSGPU_STREAM_PERSIST=$p
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_p$p -o f -- python3 $ARGR > $OUT/fetch_p$p.log 2>&1 && echo fetch $p done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_p$p -o w -- python3 $ARGR > $OUT/write_p$p.log 2>&1 && echo write $p done
done
python3 - <<PY
import sqlite3, glob
for p in (1, 0):
    for pas, c, f in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        for db in glob.glob("$OUT/%s_p%d/**/*.db" % (pas, p), recursive=True):
            con = sqlite3.connect(db)
            for name, n, mean in con.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = '%s' group by name order by avg(counter_value) desc limit 1" % c):
                print("persist=%d %s %-40s launches %d  %.1f GB per launch" % (p, c, name[:40], n, f * mean * 1024 / 1e9))
PY
