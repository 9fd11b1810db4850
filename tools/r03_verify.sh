#!/bin/bash
# The -DMIRT_VERIFY_BROAD build (variants/verify.so): every exact candidate must be flagged by the broad phase AND sit in a tile the
# tile cull kept.  Bench frame in both precisions, then the random sweeps of tools/mx_verify_sweep.py.  Every count must be 0.
export MIRT_LIB=$PWD/variants/verify.so PYTHONPATH=$PWD
for p in 32 64; do
  timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --precision $p > gpurun_out/verify_bench_$p.log 2>&1 || { tail -5 gpurun_out/verify_bench_$p.log; exit 1; }
  echo "bench frame f$p:"; grep "verify broad" gpurun_out/verify_bench_$p.log | sort | uniq -c
done
for s in 1 2 3; do
  timeout -k 10 300 python3 -u tools/mx_verify_sweep.py $s > gpurun_out/mx_sweep_$s.log 2>&1 || { tail -5 gpurun_out/mx_sweep_$s.log; exit 1; }
  echo "sweep seed $s (f32):"; grep "verify broad" gpurun_out/mx_sweep_$s.log | sort | uniq -c | tail -3; tail -1 gpurun_out/mx_sweep_$s.log
done
MIRT_SWEEP_PRECISION=64 timeout -k 10 300 python3 -u tools/mx_verify_sweep.py 4 > gpurun_out/mx_sweep_64.log 2>&1 || { tail -5 gpurun_out/mx_sweep_64.log; exit 1; }
echo "sweep seed 4 (f64):"; grep "verify broad" gpurun_out/mx_sweep_64.log | sort | uniq -c | tail -3
MIRT_SWEEP_NMIN=1100 MIRT_SWEEP_NMAX=9000 timeout -k 10 600 python3 -u tools/mx_verify_sweep.py 5 > gpurun_out/mx_sweep_big.log 2>&1 || { tail -5 gpurun_out/mx_sweep_big.log; exit 1; }
echo "sweep seed 5 (f32, 1100..9000 spheres: streamed tiles, one top tile per 32):"; grep "verify broad" gpurun_out/mx_sweep_big.log | sort | uniq -c | tail -3; tail -1 gpurun_out/mx_sweep_big.log
MIRT_SWEEP_PRECISION=64 MIRT_SWEEP_NMIN=520 MIRT_SWEEP_NMAX=6000 timeout -k 10 600 python3 -u tools/mx_verify_sweep.py 6 > gpurun_out/mx_sweep_big64.log 2>&1 || { tail -5 gpurun_out/mx_sweep_big64.log; exit 1; }
echo "sweep seed 6 (f64, 520..6000 spheres: streamed tiles):"; grep "verify broad" gpurun_out/mx_sweep_big64.log | sort | uniq -c | tail -3; tail -1 gpurun_out/mx_sweep_big64.log
