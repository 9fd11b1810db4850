#!/bin/bash
# Same-box A/B of library builds with the work counters: tools/ab_diag.sh "a b ..." [bench args...]  (variants/<name>.so; two interleaved rounds)
names=$1; shift
for round in 1 2; do
  for n in $names; do
    MIRT_LIB=$PWD/variants/$n.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$n', 'kernel_ms', r['kernel_ms_avg'], 'step_ms', d['ms_per_step'], 'tiles/iter', r.get('tiles_evaluated_per_wave_iteration'), 'seg', d['diagnostics']['segments_per_step'], 'util', d['diagnostics']['lane_utilisation'], 'narrow/seg', d['diagnostics']['narrow_tests_per_segment'])" || exit 1
  done
done
