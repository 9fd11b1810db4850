#!/bin/bash
# The fuzz campaigns of profiles/r03 (final library).txt in one go (each under its own time limit; stop at the first failure).
export PYTHONPATH=$PWD
set -o pipefail
timeout -k 10 900 python3 -u tools/fuzz_exact.py 5000 ${1:-600} > gpurun_out/fuzz_exact_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_exact_r03.log; exit 1; }
tail -2 gpurun_out/fuzz_exact_r03.log
FUZZ_MAX_N=2500 timeout -k 10 300 python3 -u tools/fuzz_exact.py 9000 40 > gpurun_out/fuzz_exact_big_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_exact_big_r03.log; exit 1; }
tail -1 gpurun_out/fuzz_exact_big_r03.log
FUZZ_IMAGES=1 timeout -k 10 500 python3 -u tools/fuzz_exact.py 20000 ${2:-1500} > gpurun_out/fuzz_exact_img_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_exact_img_r03.log; exit 1; }
tail -1 gpurun_out/fuzz_exact_img_r03.log
timeout -k 10 300 python3 -u tools/fuzz_maps.py 300 80 > gpurun_out/fuzz_maps_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_maps_r03.log; exit 1; }
tail -1 gpurun_out/fuzz_maps_r03.log
timeout -k 10 400 python3 -u tools/fuzz_radiance.py 7000 150 > gpurun_out/fuzz_radiance_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_radiance_r03.log; exit 1; }
tail -1 gpurun_out/fuzz_radiance_r03.log
timeout -k 10 300 python3 -u tools/fuzz_partition.py 100 100 > gpurun_out/fuzz_partition_r03.log 2>&1 || { tail -5 gpurun_out/fuzz_partition_r03.log; exit 1; }
tail -1 gpurun_out/fuzz_partition_r03.log
echo "fuzz campaigns done (the verify sweeps are tools/r03_verify.sh)"
