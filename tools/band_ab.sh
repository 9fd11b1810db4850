#!/bin/bash
# tools/band_ab.sh "a b ..." : the 1/8-band probe (tools/inflight_probe.py 8 40) for variants/<name>.so, one after the other on one box
for v in $1; do echo "== $v"; MIRT_LIB=$PWD/variants/$v.so timeout -k 10 200 python3 tools/inflight_probe.py 8 40 2>/dev/null || exit 1; done
