#!/bin/bash
# sweep of the traversal-suspension threshold (mirt_opts.bvh_keep, in 64ths) for the HBM-tree kernel
for k in 8 16 24 40; do
  for cfg in "--workload stress10000 --width 3840 --spp 16" "--workload stress100000 --width 3840 --spp 16"; do
    timeout -k 10 300 python bench.py --accel bvh --bvh-keep $k --no-cpu-baseline --steps 2 $cfg | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('keep $k', d['config']['workload'][:14], d['value'],'Mrays/s', d['ms_per_step'],'ms')" || exit 1
  done
done
