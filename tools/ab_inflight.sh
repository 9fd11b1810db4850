#!/bin/bash
# A/B of library builds (variants/<name>.so) on the full bench frame and on one rank's band of an 8-way split, 1-4 frames in flight.
for round in 1 2; do for v in "$@"; do echo "== $v"; MIRT_LIB=$PWD/variants/$v.so PYTHONPATH=$PWD timeout -k 10 300 python3 tools/inflight_probe.py 8 30 2>/dev/null | tail -3 | cut -c1-110; MIRT_LIB=$PWD/variants/$v.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full frame', d['ms_per_step'], 'three in flight', d['frames_in_flight_3']['ms_per_frame'])"; done; done
