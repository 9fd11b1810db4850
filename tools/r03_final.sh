#!/bin/bash
# Round 3, final library: full GPU test log, the bench lines, the profile set, the stress runs.  Outputs under gpurun_out/; tools/r03_collect.sh copies them into profiles/.
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/t_final.log 2>&1; rc=$?; tail -3 gpurun_out/t_final.log; [ $rc -eq 0 ] || exit $rc
bash tools/r03_benches.sh || exit 1
bash tools/r03_profiles.sh || exit 1
bash tools/r03_stress_runs.sh > gpurun_out/r03_stress_runs.txt 2>&1 || { tail -5 gpurun_out/r03_stress_runs.txt; exit 1; }
cat gpurun_out/r03_stress_runs.txt
