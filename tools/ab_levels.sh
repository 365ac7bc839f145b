#!/bin/bash
# A/B of an environment switch over the small configurations, same box: bash tools/ab_levels.sh MG_SMALLNET "3:8 4:32 5:64 6:6 7:6"
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=${1:-MG_SMALLNET}
CASES=${2:-"3:8 4:32 5:64 6:6 7:6"}
VALS=${3:-"0 1 0 1"}   # the values to alternate, e.g. "0 32 0 32"
for c in $CASES; do
  L=${c%%:*}; B=${c##*:}
  for v in $VALS; do
    env $VAR=$v python3 $R/bench.py --level $L --batch $B --steps 200 --warmup 60 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('L$L bs$B $VAR=$v  %.3f ms  %.0f img/s  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
  done
done
