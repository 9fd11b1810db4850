#!/bin/bash
# sweep of samples-per-work-item (mirt_opts.chunk_samples) on the 1080p bench frame and on one rank's band of an 8-way split
for c in 2 4 8; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 --chunk $c > gpurun_out/cs.json 2> gpurun_out/cs.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/cs.json')); r=d['roofline']; print('chunk $c: K2', r['kernel_ms_avg'], 'K2b', r['resolve_kernel_ms_avg'], 'step', d['ms_per_step'], 'lane util', d['diagnostics']['lane_utilisation'])"
  PROBE_CHUNK=$c PYTHONPATH=$PWD timeout -k 10 300 python3 tools/inflight_probe.py 8 30 2>/dev/null | tail -3 | cut -c1-120
done
