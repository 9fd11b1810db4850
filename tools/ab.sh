#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh "a_base b_new ..." [bench args...]  (variants/<name>.so; two interleaved rounds)
names=$1; shift
for round in 1 2; do
  for n in $names; do
    MIRT_LIB=$PWD/variants/$n.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', 'kernel_ms', d['roofline']['kernel_ms_avg'], 'step_ms', d['ms_per_step'], d['roofline']['kernel'])" || exit 1
  done
done
