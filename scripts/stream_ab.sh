#!/bin/bash
# A/B of the streamed solver's launch forms on bench.py's realistic mix (same box, alternating runs)
# usage: stream_ab.sh [bins] ["qtail values"]
BINS=${1:-4096}
run() {
  python bench.py --workload realistic --bins $BINS --steps 2 --warmup 1 --no-cpu | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d.get('realistic_mix', d)
print('  %.0f bins/s  %.1f ms/step' % (r['value'], r['ms_per_step']))"
}
for i in 1 2; do
  echo "one workgroup per bin"; SOSGPU_STREAM_PERSIST=0 run
  for q in ${2:--1}; do
    echo "persistent, q_tail=$q"; SOSGPU_STREAM_PERSIST=1 SOSGPU_STREAM_QTAIL=$q run
  done
done
