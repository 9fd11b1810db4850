#!/bin/bash
# How many paths the late waves hand to the continuation launch: variants/phase.so prints the count (diagnostic build).
# The stress frame of the bench, and the frame of tests/test_gpu_broad.py::test_continuation_pass_frame_bands_and_passes_are_bit_identical.
export PYTHONPATH=$PWD MIRT_LIB=$PWD/variants/phase.so
timeout -k 10 200 python3 bench.py --workload stress10000 --width 3840 --spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "continuation launch" | tail -1 || exit 1
timeout -k 10 200 python3 - <<'P' 2>&1 | grep -E "continuation launch|segments" || exit 1
from raytracer_amd import abi, presets
sc, cam = presets.stress_scene(1500, seed=12), presets.stress_camera(2560, 16, 50)
cam.render_arrays(sc, precision=abi.F32, seed=9, want_rgb8=False)
print("test frame: segments", cam.last_stats["segments"])
P
