#!/bin/bash
# BVH path on the bench workloads (one line each)
for cfg in "--accel bvh" "--accel bvh --workload stress10000 --width 3840 --spp 16" "--accel bvh --workload stress100000 --width 3840 --spp 16"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 $cfg | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'], d['value'],'Mrays/s', d['ms_per_step'],'ms', d['diagnostics'])" || exit 1
done
