#!/bin/bash
# Same-box A/B of library builds on the resolve kernel K2b (mirt_stats.resolve_ms): tools/ab_resolve.sh "a b ..." [bench args...]
names=$1; shift
for round in 1 2; do
  for n in $names; do
    MIRT_LIB=$PWD/variants/$n.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', 'K2 ms', d['roofline']['kernel_ms_avg'], 'K2b ms', d['roofline']['resolve_kernel_ms_avg'], 'frame ms', d['ms_per_step'])" || exit 1
  done
done
