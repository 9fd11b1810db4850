#!/bin/bash
# Long fuzz campaigns for the record (final library).  Each step writes its log as it goes.
export PYTHONPATH=$PWD
set -o pipefail
timeout -k 10 900 python3 -u tools/fuzz_maps.py 4000 ${1:-1500} > gpurun_out/soak_r03_maps.log 2>&1 || { tail -5 gpurun_out/soak_r03_maps.log; exit 1; }
tail -1 gpurun_out/soak_r03_maps.log
timeout -k 10 900 python3 -u tools/fuzz_radiance.py 30000 ${2:-2000} > gpurun_out/soak_r03_radiance.log 2>&1 || { tail -5 gpurun_out/soak_r03_radiance.log; exit 1; }
tail -1 gpurun_out/soak_r03_radiance.log
timeout -k 10 900 python3 -u tools/fuzz_partition.py 7000 ${3:-1500} > gpurun_out/soak_r03_partition.log 2>&1 || { tail -5 gpurun_out/soak_r03_partition.log; exit 1; }
tail -1 gpurun_out/soak_r03_partition.log
FUZZ_MAX_N=2500 timeout -k 10 900 python3 -u tools/fuzz_exact.py 40000 ${4:-400} > gpurun_out/soak_r03_exact_big.log 2>&1 || { tail -5 gpurun_out/soak_r03_exact_big.log; exit 1; }
tail -1 gpurun_out/soak_r03_exact_big.log
