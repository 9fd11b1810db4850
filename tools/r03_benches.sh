#!/bin/bash
# The bench lines kept under profiles/ (r03_*_bench.json), one after the other.
set -o pipefail
b() { out=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/$out.json 2> gpurun_out/$out.err || { tail -5 gpurun_out/$out.err; exit 1; }; echo "done $out"; }
b r03_mx_bench
b r03_stress10k_bench --workload stress10000 --width 3840 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline
b r03_textured_bench --workload textured --width 3840 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline
b r03_node_bench --node --no-cpu-baseline
