#!/bin/bash
# sweep of samples-per-work-item (mirt_opts.chunk_samples) on the 1080p bench and on a small frame
for c in 1 2 4; do
  for cfg in "--width 1920 --spp 64" "--width 640 --spp 8" "--width 680 --spp 64"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --chunk $c $cfg > gpurun_out/cs.json 2> gpurun_out/cs.err || exit 1
    python -c "import json; d=json.load(open('gpurun_out/cs.json')); print('chunk $c', '$cfg', 'ms', d['roofline']['kernel_ms_avg'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'util', d['diagnostics']['lane_utilisation'])"
  done
done
