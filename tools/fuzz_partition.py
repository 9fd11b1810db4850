"""Fuzz of the work decomposition: random image sizes, sample counts, rank counts, strip heights, pass cuts, queue-order
and work-item knobs - the frame assembled from bands and progressive passes must be the one-call frame bit for bit
(linear radiance, bytes, segment count), in both precisions and through every front end.
usage: PYTHONPATH=. python tools/fuzz_partition.py [first_seed] [n_cases]      (FUZZ_MAX_W / FUZZ_MAX_SPP: larger frames)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

from raytracer_amd import abi, dist, presets

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = abi.load()
scenes = {"balls": presets.bouncing_balls_scene(1), "mars": presets.mars_texture_scene()}
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    knob_q = int(rng.choice([1, 2, 4, 8])) if rng.random() < 0.5 else 0   # mirt_opts.chunk_samples
    knob_g = int(rng.choice([1, 2, 4, 16, 64])) if rng.random() < 0.5 else 0  # mirt_opts.group_chunks
    name = "balls" if rng.random() < 0.8 else "mars"
    sc = scenes[name]
    W, spp, depth = int(rng.integers(1, int(os.environ.get("FUZZ_MAX_W", "260")))), int(rng.integers(1, int(os.environ.get("FUZZ_MAX_SPP", "40")))), int(rng.integers(1, 12))
    cam = (presets.bouncing_balls_camera if name == "balls" else presets.default_camera)(W, spp, depth)
    cam.chunk_samples, cam.group_chunks = knob_q, knob_g
    H = cam.height
    prec = abi.F64 if rng.random() < 0.3 else abi.F32
    real, rsz = (torch.float64, 8) if prec == abi.F64 else (torch.float32, 4)
    flags = int(rng.choice([0, abi.FLAG_VALU_BROAD] + ([abi.FLAG_BVH] if prec == abi.F32 else [])))
    seed = seed0 + case
    ds = sc.device_scene(0)
    ref = torch.zeros((H, W, 3), dtype=real, device="cuda")
    ref8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    st = abi.MirtStats()
    o = cam.opts(precision=prec, seed=seed, flags=flags)
    abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), ref.data_ptr(), ref8.data_ptr(), None, C.byref(st)))
    seg_ref = st.segments
    n, S = int(rng.integers(1, 9)), int(rng.choice([1, 3, 8, 16]))
    g = lib.mirt_pass_granularity(C.byref(cam.pod), C.byref(o))
    units = (spp + g - 1) // g
    cuts = sorted(set([0, spp] + [int(c) * g for c in rng.integers(0, units + 1, int(rng.integers(0, 4))) if 0 < int(c) * g < spp]))
    rows0 = dist.part_rows(H, S, 0, n)
    bands = torch.zeros((n, rows0, W, 3), dtype=real, device="cuda")
    bands8 = torch.zeros((n, rows0, W, 3), dtype=torch.uint8, device="cuda")
    acc = torch.full((n, rows0, W, 3), float("nan"), dtype=real, device="cuda")
    seg = 0
    for r in range(n):
        ob = cam.opts(precision=prec, seed=seed, strip_rows=S, part_index=r, part_count=n, flags=flags | abi.FLAG_COMPACT_ROWS)
        for a, b in zip(cuts[:-1], cuts[1:]):
            st = abi.MirtStats()
            abi.check(lib.mirt_render_pass_device(ds.handle, C.byref(cam.pod), C.byref(ob), a, b - a, acc[r].data_ptr(), bands[r].data_ptr(), bands8[r].data_ptr(), None, C.byref(st)))
            seg += st.segments
    frame = torch.empty((H, W, 3), dtype=real, device="cuda")
    frame8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    abi.check(lib.mirt_deinterleave_device(bands.data_ptr(), bands[0].numel() * rsz, n, S, W, H, 3 * rsz, frame.data_ptr(), None))
    abi.check(lib.mirt_deinterleave_device(bands8.data_ptr(), bands8[0].numel(), n, S, W, H, 3, frame8.data_ptr(), None))
    torch.cuda.synchronize()
    ok = torch.equal(frame, ref) and torch.equal(frame8, ref8) and seg == seg_ref
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: lin {bool(torch.equal(frame, ref))} bytes {bool(torch.equal(frame8, ref8))} segments {seg} vs {seg_ref}", flush=True)
    print(f"case {seed}: {name} {W}x{H}x{spp} depth {depth} f{prec} flags {flags} parts {n} strip {S} cuts {cuts} Q {g} knobs {knob_q}/{knob_g}", flush=True)
print(f"cases {n_cases}  mismatching frames: {bad}")
sys.exit(1 if bad else 0)
