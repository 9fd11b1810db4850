#!/bin/bash
# per-kernel durations of a band launch, row order (64) against the ordered queue (0)
root=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp && cd $root
for f in 64 0; do
  rm -rf gpurun_out/bk_$f
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bk_$f -o k -- python3 tools/band_kernels.py $f 20 > gpurun_out/bk_$f.log 2>&1 || { tail -5 gpurun_out/bk_$f.log; exit 1; }
  echo "== flags $f"; find gpurun_out/bk_$f -name "*kernel_stats.csv" -exec cat {} \; | cut -d, -f1-6 | cut -c1-150
done
