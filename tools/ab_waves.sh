for v in a_w16 b_w8; do echo "== $v"; MIRT_LIB=$PWD/variants/$v.so PYTHONPATH=$PWD timeout -k 10 300 python3 tools/inflight_probe.py 8 40 2>/dev/null | tail -3; MIRT_LIB=$PWD/variants/$v.so timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full frame', d['ms_per_step'])"; done
