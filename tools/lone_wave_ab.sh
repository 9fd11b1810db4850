#!/bin/bash
# tools/lone_wave_ab.sh "a b ..." : tools/lone_wave_probe.py for variants/<name>.so, one after the other on one box
for v in $1; do echo "== $v"; MIRT_LIB=$PWD/variants/$v.so timeout -k 10 120 python3 tools/lone_wave_probe.py 2>/dev/null | sed 's/.*px = lanes/ lanes/' | cut -c1-150 || exit 1; done
