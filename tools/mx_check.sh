#!/bin/bash
# GPU tests, then the bench with the matrix-pipe and the VALU broad phase, then the -DMIRT_VERIFY_BROAD build (variants/verify.so)
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_mx.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/t_mx.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/b_mx.log 2>&1 || { tail -5 gpurun_out/b_mx.log; exit 1; }
tail -1 gpurun_out/b_mx.log
timeout -k 10 120 python bench.py --accel valu --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/b_valu.log 2>&1 || { tail -5 gpurun_out/b_valu.log; exit 1; }
tail -1 gpurun_out/b_valu.log
if [ -f variants/verify.so ]; then
  MIRT_LIB=$PWD/variants/verify.so timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/b_verify.log 2>&1 || { tail -5 gpurun_out/b_verify.log; exit 1; }
  grep "verify broad" gpurun_out/b_verify.log | sort | uniq -c
fi
