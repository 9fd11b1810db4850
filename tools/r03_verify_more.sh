#!/bin/bash
# More seeds of the -DMIRT_VERIFY_BROAD sweeps (variants/verify.so): f32 and f64, scenes of 24..700 and of 520..9000 spheres.  Every count must be 0.
export MIRT_LIB=$PWD/variants/verify.so PYTHONPATH=$PWD
tot=0
for s in 7 8 9 10 11 12 13 14; do
  timeout -k 10 300 python3 -u tools/mx_verify_sweep.py $s > gpurun_out/vm_$s.log 2>&1 || { tail -5 gpurun_out/vm_$s.log; exit 1; }
  echo "seed $s f32 24..700:   $(grep 'verify broad' gpurun_out/vm_$s.log | grep -v ': 0$' | wc -l) launches with misses of $(grep -c 'verify broad' gpurun_out/vm_$s.log)"
done
for s in 15 16 17 18; do
  MIRT_SWEEP_PRECISION=64 timeout -k 10 300 python3 -u tools/mx_verify_sweep.py $s > gpurun_out/vm_$s.log 2>&1 || { tail -5 gpurun_out/vm_$s.log; exit 1; }
  echo "seed $s f64 24..700:   $(grep 'verify broad' gpurun_out/vm_$s.log | grep -v ': 0$' | wc -l) launches with misses of $(grep -c 'verify broad' gpurun_out/vm_$s.log)"
done
for s in 19 20 21; do
  MIRT_SWEEP_NMIN=1100 MIRT_SWEEP_NMAX=9000 timeout -k 10 600 python3 -u tools/mx_verify_sweep.py $s > gpurun_out/vm_$s.log 2>&1 || { tail -5 gpurun_out/vm_$s.log; exit 1; }
  echo "seed $s f32 1100..9000: $(grep 'verify broad' gpurun_out/vm_$s.log | grep -v ': 0$' | wc -l) launches with misses of $(grep -c 'verify broad' gpurun_out/vm_$s.log)"
done
for s in 22 23; do
  MIRT_SWEEP_PRECISION=64 MIRT_SWEEP_NMIN=520 MIRT_SWEEP_NMAX=6000 timeout -k 10 600 python3 -u tools/mx_verify_sweep.py $s > gpurun_out/vm_$s.log 2>&1 || { tail -5 gpurun_out/vm_$s.log; exit 1; }
  echo "seed $s f64 520..6000:  $(grep 'verify broad' gpurun_out/vm_$s.log | grep -v ': 0$' | wc -l) launches with misses of $(grep -c 'verify broad' gpurun_out/vm_$s.log)"
done
