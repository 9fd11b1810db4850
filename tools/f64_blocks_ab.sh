#!/bin/bash
# f64 matrix-pipe kernel: register budget for 2 / 3 / 4 four-wave blocks per CU (variants/f64b2.so, product, variants/f64b4.so)
for v in f64b2 "" f64b4; do
  if [ -n "$v" ]; then export MIRT_LIB=$PWD/variants/$v.so; else unset MIRT_LIB; fi
  echo "== ${v:-product (3)}"
  PYTHONPATH=$PWD timeout -k 10 200 python3 tools/cull_probe.py 2>&1 | grep "f64 tile cull" || exit 1
  PYTHONPATH=$PWD timeout -k 10 200 python3 tools/cull_probe.py stress10000 3840 16 2>&1 | grep "f64 tile cull" || exit 1
done
