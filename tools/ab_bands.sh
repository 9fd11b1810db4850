#!/bin/bash
# A/B the kernel variants in variants/*.so on one rank's share of the bench frame (tools/band_times.py; default 8 parts).
for f in variants/*.so; do
  echo "== $f"
  MIRT_LIB=$PWD/$f PYTHONPATH=. timeout -k 10 200 python3 tools/band_times.py "${@:-8}" 2>&1 | grep strip || exit 1
done
