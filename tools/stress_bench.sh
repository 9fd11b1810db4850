#!/bin/bash
# Stress scene (10 000 spheres, 4K, 16 spp) through the three closest-hit front ends (one line each)
for cfg in "--accel brute" "--accel valu" "--accel bvh"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --workload stress10000 --width 3840 --spp 16 $cfg | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'], d['roofline']['kernel'], d['value'],'Mrays/s', d['ms_per_step'],'ms frac', d['roofline']['frac'], d['diagnostics'])" || exit 1
done
