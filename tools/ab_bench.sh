#!/bin/bash
# A/B the kernel variants in variants/*.so on one GPU: one process per variant, same workload.
for f in variants/*.so; do
  echo "== $f"
  MIRT_LIB=$PWD/$f timeout -k 10 200 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras "$@" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'],'Mrays/s', d['ms_per_step'],'ms', 'frac',d['roofline']['frac'], d['diagnostics'])" || exit 1
done
