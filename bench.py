#!/usr/bin/env python3
"""bench.py - throughput of the hot path on MI355X.

One "step" = one full render of the workload (BASELINE.json configs[2], the configuration the metric is quoted on:
the ~485-sphere bouncing-balls scene, 1920x1080, 64 spp, depth 50) with the scene already resident in HBM.
N GPUs (one process per GPU, torchrun): the frame is sharded in interleaved 8-row strips, every rank renders its
band into HBM, the bands are gathered TO RANK 0 with RCCL (torch.distributed backend "nccl": grouped send/recv, every peer
over its own xGMI link; --gather allgather selects the ring all-gather instead) and scattered into frame order
by kernel K4.  Strong scaling: the frame is fixed, value = ray segments of the whole frame / time.
On several GPUs up to 3 frames (--in-flight) are in flight at a time, each on its own stream and scene handle: the
gather of one frame, and the few long paths that end it, run beside the next frame's kernel.  On one GPU the default
is one frame at a time, so that every launch has the chip to itself and its HIP-event duration is the kernel's own
(the roofline figure); the rate with 3 frames in flight is reported beside it (`frames_in_flight_3`).

Prints ONE JSON line on rank 0 (see the task contract); extra keys: roofline, cpu_baseline, diagnostics.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)": 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz
FP64_VALU_PEAK_TFLOPS = 78.6
NOMINAL_CLOCK_GHZ = 2.4  # the clock both nominal peaks are quoted at; under this load the chip runs at ~2.0 GHz (profiles/*_sq_pmc.txt)
SIMDS_PER_CU = 4
# Issue slots of the matrix-pipe broad phase that no schedule can avoid: per wave and tile of 32 spheres
# 4 x v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles each: the whole quadratic form of the conservative test, K = 32 over
# two chained MFMAs) and, per (ray, sphere) pair, the one v_alignbit_b32 that collects its sign (32 wave-instructions of
# 2 cycles); priced additively (bench/mfma_coexec.hip: ~4.5 VALU per MFMA hide in its shadow, so this is the conservative
# figure).  Per segment, the `5` of SURVEY 8d (a = D.D) is 3 more instructions.
MX_CYCLES_PER_WAVE_TILE = 4 * 32 + 32 * 2
MX_CYCLES_PER_WAVE_SEGMENT = 3 * 2
# The same slots priced with the overlap the hardware gives (bench/mfma_coexec.hip, profiles/r02_microbench.txt): ~4.5 VALU
# instructions issue for free in the 32-cycle shadow of each MFMA, so of the 32 per tile only 32 - 4 x 4.5 = 14 cost their
# 2 cycles.  The lower (harder) of the two bounds; both are printed.
MX_CYCLES_PER_WAVE_TILE_SHADOW = 4 * 32 + (32 - 4 * 4.5) * 2
# What the tile loop ALONE does on this chip, whatever the order of its instructions (bench/mx_tile_rate.hip, five schedules,
# 4 waves per SIMD, profiles/r03_mx_tile_rate.txt): 185-194 shader cycles per wave-tile.  v_alignbit_b32 (one per ray-sphere
# pair) issues at 3.4 cycles, not 2 (bench/valu_op_rate.hip); the 32 of a tile and the ~15 other VALU instructions fill the time
# the four MFMAs leave.  Not a bound - a measured ceiling of this formulation, printed beside the two bounds.
MX_CYCLES_PER_WAVE_TILE_MEASURED = 190
PROFILE_TAGS = {("balls1080p", 1): "r03_mx", ("balls1080p", 0): "r03_valu", ("stress10000", 1): "r03_stress10k", ("stress10000", 0): "r03_stress10k_valu",
                ("textured", 5): "r03_textured"}
PROFILE_FALLBACK = {"r03_": "r02_"}  # a workload not re-profiled this round falls back to the last round's committed passes


def data_note(workload):
    """`data` of the result line, per workload (the scenes are seeded restatements of what the reference builds or of what
    BASELINE.json's configs describe; nothing is downloaded)."""
    if workload == "balls1080p":
        return "synthetic (seeded restatement of the reference's bouncing_balls_scene, main.rs:151-229; reference textures pre-decoded)"
    if workload == "mars":
        return "synthetic (the reference's mars_texture_scene, main.rs:136-149; reference texture pre-decoded)"
    if workload == "textured":
        return "synthetic (BASELINE configs[3]: two image-textured spheres + checker ground, built from the reference's texture files, pre-decoded)"
    if workload.startswith("stress"):
        return "synthetic (BASELINE configs[4]: seeded uniform field of static spheres, 70/20/10 % Lambertian/metal/glass; no reference counterpart)"
    return "synthetic"


def profile_traffic(tag, kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes profiles/<tag>_hbm_traffic.txt (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, tools/profile.sh).  Calibration (MI355X_MICROARCH.md: FETCH_SIZE halves WIDE
    streaming reads, other patterns are to be calibrated on a known byte count): resolve_kernel reads the partial sums K2
    wrote exactly once - WRITE_SIZE of K2 771.4 MiB, FETCH_SIZE of resolve 739.6 MiB raw - so for the access patterns of
    these kernels the raw FETCH_SIZE is the byte count (factor 1.04, not 2); applied as 1."""
    path = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.txt")
    if not os.path.exists(path):
        for new, old in PROFILE_FALLBACK.items():
            if tag.startswith(new) and os.path.exists(os.path.join(ROOT, "profiles", f"{old}{tag[len(new):]}_hbm_traffic.txt")):
                return profile_traffic(old + tag[len(new):], kernel)
        return None
    vals = {}
    for line in open(path):
        if kernel in line and " raw " in line:
            f = line.split()
            c = "FETCH_SIZE" if "FETCH_SIZE" in f else "WRITE_SIZE" if "WRITE_SIZE" in f else None
            if c:
                vals[c] = float(f[f.index("raw") + 1]) * 1024.0  # the last launch of the run
    if len(vals) != 2:
        return None
    return {"bytes": vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "fetch": vals["FETCH_SIZE"], "write": vals["WRITE_SIZE"], "file": f"profiles/{tag}_hbm_traffic.txt"}


def profile_phase_split(tag):
    """Shares of the render kernel's wave-cycles per phase from the committed pass profiles/<tag>_phase_split.txt (the
    -DMIRT_PHASE_TIMING build, tools/phase_split.sh): {'queue': .., 'raygen': .., 'broad': .., 'narrow': .., 'shade': .., 'other': ..}."""
    path = os.path.join(ROOT, "profiles", f"{tag}_phase_split.txt")
    if not os.path.exists(path):
        return None
    import re

    for line in open(path):
        if "wave-cycles:" in line:
            d = {k: float(v) / 100.0 for k, v in re.findall(r"(queue|raygen|broad|narrow|shade|other) ([0-9.]+)%", line)}
            if d:
                d["file"] = f"profiles/{tag}_phase_split.txt"
                return d
    return None


def profile_valu_rate(tag):
    """VALU wave-instructions per SIMD-cycle from the committed SQ pass profiles/<tag>_sq_pmc.txt (tools/pmc_report.py line)."""
    path = os.path.join(ROOT, "profiles", f"{tag}_sq_pmc.txt")
    if not os.path.exists(path):
        for new, old in PROFILE_FALLBACK.items():
            if tag.startswith(new) and os.path.exists(os.path.join(ROOT, "profiles", f"{old}{tag[len(new):]}_sq_pmc.txt")):
                return profile_valu_rate(old + tag[len(new):])
        return None
    import re

    for line in open(path):
        m = re.search(r"VALU wave-instructions per SIMD-cycle = ([0-9.]+)", line)
        if m:
            return float(m.group(1))
    return None


def build_workload(name, width, spp, depth):
    from raytracer_amd import presets

    if name == "balls1080p":
        return presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(width, spp, depth), "bouncing_balls(seed=1)"
    if name == "mars":
        return presets.mars_texture_scene(), presets.default_camera(width, spp, depth), "mars_texture_scene"
    if name == "textured":
        return presets.textured_scene(), presets.textured_camera(width, spp, depth), "textured_scene"
    if name.startswith("stress"):
        n = int(name[6:] or 10000)
        return presets.stress_scene(n, 1), presets.stress_camera(width, spp, depth), f"stress({n})"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline_c1(seed):
    """BASELINE.json configs[0], the reference's own default render (main.rs:30-56,126-127): mars_texture_scene, 400 x 225,
    100 spp, depth 50, lens 0.4 deg - the WHOLE frame on one thread (the reference is single-threaded), f64 oracle with the
    reference's BVH walk."""
    from oracle import binding as ob
    from raytracer_amd import abi, presets

    sc, cam = presets.mars_texture_scene(), presets.default_camera(400, 100, 50)
    t0 = time.perf_counter()
    _, _, seg, _ = ob.render(sc.flat, cam.pod, abi.F64, seed=seed, use_bvh=True, threads=1, want_rgb8=True)
    dt = time.perf_counter() - t0
    return {"value": round(seg / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port", "seconds_per_frame": round(dt, 3),
            "sample": f"the reference's default render in full: mars_texture_scene {cam.width}x{cam.height}x100spp depth 50, {seg} segments in {dt:.1f} s, "
                      "f64 oracle with the reference's BVH walk, single thread"}, sc, cam


def cpu_baseline(scene, cam, seed, seconds_budget=12.0):
    """The restated reference (oracle, f64, faithful BVH walk) timed on this box's host cores on a bounded sample:
    a band of rows from the middle of the same frame, sized from a 1-row probe to about `seconds_budget` seconds."""
    from oracle import binding as ob
    from raytracer_amd import abi

    H = cam.height
    mid = H // 2
    t0 = time.perf_counter()
    _, _, seg, _ = ob.render(scene.flat, cam.pod, abi.F64, seed=seed, use_bvh=True, threads=1, rows=(mid, mid + 1), want_rgb8=False)
    probe = time.perf_counter() - t0
    rows = max(1, min(H - mid, int(seconds_budget / max(probe, 1e-3))))
    t0 = time.perf_counter()
    _, _, seg, _ = ob.render(scene.flat, cam.pod, abi.F64, seed=seed, use_bvh=True, threads=1, rows=(mid, mid + rows), want_rgb8=False)
    dt1 = time.perf_counter() - t0
    single = seg / dt1 / 1e6
    threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU is 16 cores
    rows_mt = max(threads, min(H - mid, rows * threads // 2))
    t0 = time.perf_counter()
    _, _, seg_mt, _ = ob.render(scene.flat, cam.pod, abi.F64, seed=seed, use_bvh=True, threads=threads, rows=(mid, mid + rows_mt), want_rgb8=False)
    dtn = time.perf_counter() - t0
    base = {
        "value": round(single, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
        "sample": f"rows {mid}..{mid + rows} of the same {cam.width}x{H}x{cam.pod.samples_per_pixel}spp frame, f64 oracle with the reference's BVH walk, "
                  f"{seg} segments in {dt1:.1f} s, single thread (the reference is single-threaded)",
    }
    mt = {
        "value": round(seg_mt / dtn / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"rows {mid}..{mid + rows_mt}, row-parallel, {seg_mt} segments in {dtn:.1f} s",
    }
    return base, mt


def run_node(args, lib, scene, cam, scene_name, result_fd):
    """--node: the whole node from ONE process through mirt_node_render_device (what a drop-in Camera::render would call)."""
    import torch

    from raytracer_amd import abi
    from raytracer_amd.api import NodeScene

    n = args.gpus
    if n > torch.cuda.device_count():
        raise SystemExit(f"--node --gpus {n}: this box has {torch.cuda.device_count()} GPU(s)")
    node = NodeScene(scene.flat, list(range(n)))
    H, W = cam.height, cam.width
    real = torch.float32 if args.precision == 32 else torch.float64
    frame = torch.zeros((H, W, 3), dtype=real, device="cuda:0")
    accel_flags = {"brute": 0, "valu": abi.FLAG_VALU_BROAD, "bvh": abi.FLAG_BVH}[args.accel]
    cam.chunk_samples, cam.group_chunks = args.chunk, args.group_chunks
    o = cam.opts(precision=args.precision, seed=args.seed, flags=accel_flags, part_count=0, bvh_keep=args.bvh_keep)
    kinds = {"to_root": abi.GATHER_TO_ROOT, "allgather": abi.GATHER_ALLGATHER}

    def run(kind, steps):
        sts = []
        t0 = time.perf_counter()
        for _ in range(steps):
            st = abi.MirtNodeStats()
            abi.check(lib.mirt_node_render_device(node.handle, C.byref(cam.pod), C.byref(o), kinds[kind], frame.data_ptr(), None, C.byref(st)))
            sts.append(st)
        return time.perf_counter() - t0, sts

    run(args.gather, max(1, args.warmup) + 1)  # set-up (communicators, buffers, operands) and warm-up, untimed
    elapsed, sts = run(args.gather, args.steps)
    other = "allgather" if args.gather == "to_root" else "to_root"
    run(other, 1)
    el2, sts2 = run(other, max(2, args.steps // 2))
    seg = sum(s_.segments for s_ in sts)

    def view(sl):
        k = len(sl)
        return {"gather_ms": round(sum(s_.gather_ms for s_ in sl) / k, 3), "scatter_ms": round(sum(s_.scatter_ms for s_ in sl) / k, 3),
                "band_kernel_ms_per_device": [round(sum(s_.band_kernel_ms[i] for s_ in sl) / k, 3) for i in range(n)],
                "call_ms": round(sum(s_.wall_ms for s_ in sl) / k, 3)}

    out = {
        "metric": "Mrays/s at 1080p (ray segment = one closest-hit query)", "value": round(seg / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32" if args.precision == 32 else "f64", "data": data_note(args.workload),
        "config": {"workload": f"{scene_name}: {scene.flat.n_spheres} spheres, {W}x{H}, {cam.pod.samples_per_pixel} spp, depth {cam.pod.max_bounce_depth}",
                   "parallelism": f"one process, {n} GPU(s): mirt_node_render_device (scene replicas, interleaved 8-row strips, RCCL {args.gather} inside libmirt.so, K4 on device 0)",
                   "frames_in_flight": 1, "seed": args.seed},
        "multi_gpu": {"rccl_ranks": n, "single_process": True, "exchange": args.gather, **view(sts), "other_exchange": {"kind": other, "ms_per_step": round(el2 / len(sts2) * 1e3, 3), **view(sts2)}},
    }
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    node.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="balls1080p")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--precision", type=int, default=32, choices=[32, 64])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--accel", default="brute", choices=["brute", "valu", "bvh"],
                    help="closest-hit method: brute = the brute-force tile loop (the metric; broad phase on the matrix pipe where the scene allows it), "
                         "valu = the same loop with the broad phase kept on the vector ALU, bvh = the bounding-ball tree")
    ap.add_argument("--chunk", type=int, default=0, help="mirt_opts.chunk_samples: lower bound of the samples per work item (0 = the library's choice)")
    ap.add_argument("--group-chunks", type=int, default=0, help="mirt_opts.group_chunks (image-neutral tuning; 0 = automatic)")
    ap.add_argument("--bvh-keep", type=int, default=0, help="mirt_opts.bvh_keep (image-neutral tuning of the tree traversal; 0 = automatic)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra launches (valu_broad, accel_bvh): profiling runs want one kernel")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="frames in flight (ring of scene handles and streams); 1 = one frame at a time; 0 = 1 on one GPU "
                         "(each launch alone on the chip: the kernel duration the roofline is priced with), 3 on several")
    ap.add_argument("--node", action="store_true",
                    help="single process: drive --gpus N devices through the library's own multi-GPU entry point (mirt_node_render_device: scene replicas, "
                         "RCCL exchange inside libmirt.so) instead of one torch.distributed rank per GPU; prints one line of the same shape")
    ap.add_argument("--gather", default="to_root", choices=["to_root", "allgather"],
                    help="the exchange of the bands: to_root = every rank sends its band to rank 0 (grouped send/recv, each peer over its own xGMI link), "
                         "allgather = ring all-gather (N x the bytes).  --node: the one that is timed (the other is measured beside it)")
    ap.add_argument("--force-dist", action="store_true", help="run the gather path with a process group of one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets N ranks rehearse the N>1 path on ONE GPU")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: librccl prints a version banner to fd 1 when the process group is made,
    # so everything written to fd 1 before the result goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from raytracer_amd import abi, dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    local_rank %= torch.cuda.device_count()  # rehearsal on a one-GPU box: all ranks share device 0 (gloo only)
    torch.cuda.set_device(local_rank)
    td = None
    use_dist = world > 1 or args.force_dist  # --force-dist: the gather path with a process group of one (rehearsal on a one-GPU box)
    if use_dist:
        import torch.distributed as td

        if args.force_dist and "RANK" not in os.environ:
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29533"), "RANK": "0", "WORLD_SIZE": "1"})
        if args.backend == "nccl":
            pg_opts = None
            try:  # the gathers run beside the next frame's kernel: let their blocks have the first CUs that come free
                pg_opts = td.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                pass
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
        else:
            td.init_process_group(args.backend)

    lib = abi.load()
    scene, cam, scene_name = build_workload(args.workload, args.width, args.spp, args.depth)
    if args.node:
        return run_node(args, lib, scene, cam, scene_name, result_fd)
    ds = scene.device_scene(local_rank)
    H, W, S = cam.height, cam.width, dist.DEFAULT_STRIP_ROWS
    real = torch.float32 if args.precision == 32 else torch.float64
    rows0 = dist.part_rows(H, S, 0, world)
    accel_flags = {"brute": 0, "valu": abi.FLAG_VALU_BROAD, "bvh": abi.FLAG_BVH}[args.accel]
    cam.chunk_samples, cam.group_chunks = args.chunk, args.group_chunks
    opts = cam.opts(precision=args.precision, seed=args.seed, strip_rows=S, part_index=rank, part_count=world,
                    flags=abi.FLAG_COMPACT_ROWS | abi.FLAG_DEFER_STATS | accel_flags, bvh_keep=args.bvh_keep)
    kind = lib.mirt_closest_hit_kind(ds.handle, C.byref(cam.pod), C.byref(opts))  # 0 valu, 1 matrix pipe, 2 tree, 3 f64, 4 f64 with the matrix-pipe broad phase
    abi.check(min(kind, 0))
    stream = torch.cuda.current_stream().cuda_stream
    elem = 3 * (4 if args.precision == 32 else 8)

    # Frames in flight.  A frame ends with a few long paths (glass, up to the depth limit) on a few CUs while the rest of
    # the chip has nothing left to pull from the queue, and (N > 1) with the gather.  Each slot of the ring owns a scene
    # handle (work queue + partial sums), a stream and its buffers, so the next frame's blocks take the CUs as they come
    # free and the RCCL gather of frame f runs beside the kernel of frame f + 1.  Every frame is rendered, gathered and
    # scattered in full; the statistics of a slot are collected (mirt_render_wait) before the slot is used again.
    from raytracer_amd.api import DeviceScene

    class Slot:
        pass

    depth = args.in_flight if args.in_flight > 0 else (3 if use_dist else 1)
    ring = []
    for i in range(depth):
        sl = Slot()
        sl.ds = ds if i == 0 else DeviceScene(scene.flat, local_rank)
        sl.stream = torch.cuda.Stream() if depth > 1 else torch.cuda.current_stream()
        sl.band = torch.zeros((rows0, W, 3), dtype=real, device="cuda")
        sl.band8 = torch.zeros((rows0, W, 3), dtype=torch.uint8, device="cuda")
        # gather-to-root: only rank 0 holds the gathered bands
        sl.allb = torch.zeros((world * rows0, W, 3), dtype=real, device="cuda") if use_dist and (rank == 0 or args.gather == "allgather") else None
        sl.frame = torch.zeros((H, W, 3), dtype=real, device="cuda") if rank == 0 else None
        sl.busy = False
        # gloo rehearsal (N ranks on one GPU): the gather goes through pinned host copies made on the frame's own stream - torch's
        # gloo backend must never be handed a device tensor here (tools/gloo_probe.py: its device staging deadlocks while the
        # producing stream is busy, flushed or not; host tensors are fine)
        host_gather = use_dist and args.backend != "nccl"
        sl.hband = torch.empty((rows0, W, 3), dtype=real, pin_memory=True) if host_gather else None
        sl.hall = torch.empty((world * rows0, W, 3), dtype=real, pin_memory=True) if host_gather and (rank == 0 or args.gather == "allgather") else None
        sl.copied = torch.cuda.Event() if host_gather else None
        sl.gather_pending = False
        ring.append(sl)
    band, band8 = ring[0].band, ring[0].band8
    torch.cuda.synchronize()
    stats = []
    gather_events = []  # (before, after) of every gather, on the slot's stream

    def collect(sl):
        if sl.busy:
            st = abi.MirtStats()
            abi.check(lib.mirt_render_wait(sl.ds.handle, C.byref(st)))
            sl.busy = False
            stats.append(st)
        if sl.gather_pending:
            # rehearsal: the frame's band has reached pinned memory behind its kernels; gather on the host, send the frame back.
            # It happens when the slot is recycled (or at the fence), so the frames of the other slots stay in flight meanwhile.
            sl.copied.synchronize()
            exchange(sl.hall, sl.hband)
            if sl.allb is not None:
                with torch.cuda.stream(sl.stream):
                    sl.allb.copy_(sl.hall, non_blocking=True)
            if rank == 0:
                abi.check(lib.mirt_deinterleave_device(sl.allb.data_ptr(), sl.band.numel() * sl.band.element_size(), world, S, W, H, elem, sl.frame.data_ptr(), sl.stream.cuda_stream))
            sl.gather_pending = False

    def exchange(all_bands, my_band):
        """The one exchange step of a frame, on torch's current stream (device tensors: RCCL) or on the host (gloo rehearsal)."""
        if args.gather == "allgather":
            dist.gather_bands_into(all_bands, my_band)
        else:
            dist.gather_bands_to_root_into(all_bands, my_band, rank, world)

    def step(f):
        sl = ring[f % depth]
        collect(sl)
        q = sl.stream.cuda_stream
        abi.check(lib.mirt_render_device(sl.ds.handle, C.byref(cam.pod), C.byref(opts), sl.band.data_ptr(), sl.band8.data_ptr(), q, None))
        sl.busy = True
        if use_dist and args.backend != "nccl":
            with torch.cuda.stream(sl.stream):
                sl.hband.copy_(sl.band, non_blocking=True)
                sl.copied.record()
            sl.gather_pending = True
        elif use_dist:
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sl.stream):
                g0.record()
                exchange(sl.allb, sl.band)  # RCCL over xGMI: the one exchange step
                g1.record()
            gather_events.append((g0, g1))
            if rank == 0:
                abi.check(lib.mirt_deinterleave_device(sl.allb.data_ptr(), sl.band.numel() * sl.band.element_size(), world, S, W, H, elem, sl.frame.data_ptr(), q))
        else:
            abi.check(lib.mirt_deinterleave_device(sl.band.data_ptr(), sl.band.numel() * sl.band.element_size(), 1, S, W, H, elem, sl.frame.data_ptr(), q))

    def fence():
        for sl in ring:
            collect(sl)
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    for f in range(depth):  # set-up, untimed: every slot's first launch allocates its partial sums and uploads its operands
        step(f)
    fence()
    for f in range(args.warmup):
        step(f)
    fence()
    stats.clear()
    gather_events.clear()
    t0 = time.perf_counter()
    for f in range(args.steps):
        step(f)
    fence()
    elapsed = time.perf_counter() - t0
    assert len(stats) == args.steps
    if rank == 0 and depth > 1 and args.steps >= 2 and not torch.equal(ring[0].frame, ring[1].frame):
        raise SystemExit("frames of two ring slots differ")
    opts.flags &= ~abi.FLAG_DEFER_STATS

    seg = sum(s.segments for s in stats)
    flops = sum(18 * s.static_tests + 27 * s.moving_tests + 5 * s.segments for s in stats)
    launch_ms = sum(s.kernel_ms for s in stats)                  # render kernel K2 + resolve kernel K2b (events around both)
    resolve_ms = sum(s.resolve_ms for s in stats)                 # K2b alone
    kern_ms = launch_ms - resolve_ms                              # K2 alone: the kernel the roofline names
    mx_tiles = sum(s.mx_tiles for s in stats)                     # (wave, tile) evaluations of the matrix-pipe kernels: what the tile cull left
    iters = sum(s.wave_iterations for s in stats)
    narrow = sum(s.narrow_tests for s in stats)
    if td is not None:
        rdev = "cuda" if args.backend == "nccl" else "cpu"  # the reductions of the report: gloo gets host tensors only
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor([seg, flops, iters, narrow], dtype=torch.float64, device=rdev)
        td.all_reduce(v, op=td.ReduceOp.SUM)
        seg_all, flops_all, iters_all, narrow_all = (float(x) for x in v.tolist())
        # per-rank view for reading a scaling record: this rank's band kernel and its gather (event time on the slot's stream:
        # with frames in flight it includes waiting for the slowest rank's band, which is what a gather costs a rank)
        mine = torch.tensor([kern_ms / max(1, len(stats)), sum(a.elapsed_time(b) for a, b in gather_events) / max(1, len(gather_events)), float(seg)],
                            dtype=torch.float64, device=rdev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(per_rank, mine)
        per_rank = [t_.tolist() for t_ in per_rank]
    else:
        per_rank = None
        seg_all, flops_all, iters_all, narrow_all = float(seg), float(flops), float(iters), float(narrow)

    if rank == 0:
        # rank 0's kernel: per-GPU rate.  With frames in flight the launches share the chip (each one's event time spans
        # its neighbours' too), so the rate is priced with this rank's share of the wall time instead: a lower bound.
        achieved = (flops / ((kern_ms if depth == 1 else elapsed * 1e3) * 1e-3)) / 1e12 if kern_ms > 0 else 0.0
        tiny = kind == 5
        kernel_name = ["render_kernel_f32", "render_kernel_f32_mx", "render_kernel_f32_bvh(_lds)", "render_kernel_f64<2>", "render_kernel_f64_mx", "render_kernel_f32_tiny"][kind]
        n_launch = max(1, len(stats))
        ms_avg = kern_ms / n_launch
        t_priced = (kern_ms if depth == 1 else elapsed * 1e3) * 1e-3  # s; with frames in flight the launches overlap: wall time, a lower bound
        cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        measured_peak = C.c_double(0.0)
        if lib.mirt_measure_fma_peak(local_rank, C.byref(measured_peak)) != 0:
            measured_peak.value = 0.0
        tag = PROFILE_TAGS.get((args.workload, kind)) if (W, cam.pod.samples_per_pixel, world) in ((1920, 64, 1), (3840, 16, 1), (3840, 256, 1)) else None
        traffic = profile_traffic(tag, kernel_name) if tag else None
        roof = {"kernel": kernel_name, "kernel_ms_avg": round(ms_avg, 3), "resolve_kernel_ms_avg": round(resolve_ms / n_launch, 3),
                "launch_ms_avg": round(launch_ms / n_launch, 3),
                "timing_note": "kernel_ms_avg = the render kernel K2 alone (HIP events before K2 and between K2 and K2b on the launch stream); launch_ms_avg = K2 + the resolve "
                               "kernel K2b, which is what a frame costs and what `value` is made of",
                "flops_per_segment": round(flops / max(seg, 1), 1),
                "algorithmic_tflops": round(achieved, 3),
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": (traffic["file"] + " (FETCH_SIZE raw + WRITE_SIZE of the last launch; calibration in bench.py::profile_traffic)") if traffic else None,
                "traffic_algorithmic": W * H * 3 * (4 if args.precision == 32 else 8),
                "traffic_note": "algorithmic = the linear RGB frame written once (+ 4 B per image-textured bounce, L2-resident); the measured bytes are the per-work-item partial "
                                "sums K2 writes for K2b to add in a fixed order (12 B x W x H x chunks): wasted traffic by the HBM yardstick, 0.25 ms of the frame, the price of "
                                "an image that does not depend on scheduling or GPU count"}
        note = ("algorithmic flops = 18/static sphere test + 27/moving sphere test + 5/segment (SURVEY.md 8d); rank 0's launches; duration = HIP events around each launch on its own stream"
                + ("" if depth == 1 else f"; {depth} frames in flight: launches overlap, priced with this rank's wall time of the timed region"))
        if kind in (1, 4):
            # the kernel whose broad phase runs on the matrix pipe is bounded by the issue slots of the SIMD's shared VALU/MFMA port
            n_static = int(round(sum(s_.static_tests for s_ in stats) / max(seg, 1)))
            n_moving = int(round(sum(s_.moving_tests for s_ in stats) / max(seg, 1)))
            slots = (n_static + n_moving + 31) // 32 * 32  # the library's layout: the sorted array padded to whole 32-sphere tiles
            tiles_all = slots // 32
            # the tile cull: a wave evaluates the top tile and then only the tiles some ray of it can touch.  The bound prices the
            # tiles that WERE evaluated (mirt_stats.mx_tiles, counted by the kernel), not the all-pairs count.
            tiles = mx_tiles / max(iters, 1) if mx_tiles else float(tiles_all)
            cyc = tiles * MX_CYCLES_PER_WAVE_TILE + MX_CYCLES_PER_WAVE_SEGMENT
            t_bound = (seg / 64.0) * cyc / (cus * SIMDS_PER_CU * NOMINAL_CLOCK_GHZ * 1e9)  # s for all timed launches of this rank
            peak = flops / t_bound / 1e12
            cyc_sh = tiles * MX_CYCLES_PER_WAVE_TILE_SHADOW + MX_CYCLES_PER_WAVE_SEGMENT
            t_bound_sh = (seg / 64.0) * cyc_sh / (cus * SIMDS_PER_CU * NOMINAL_CLOCK_GHZ * 1e9)
            roof.update({"bound": "valu+mfma issue", "achieved": round(achieved, 3), "peak": round(peak, 3), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "issue_cycles_per_wave_and_segment": round(cyc, 1), "sphere_tiles": tiles_all, "tiles_evaluated_per_wave_iteration": round(tiles, 3),
                         "pairs_evaluated_share": round(tiles / tiles_all, 4) if tiles_all else None,
                         "clock_ghz": NOMINAL_CLOCK_GHZ, "simds": cus * SIMDS_PER_CU,
                         "bound_ms_per_launch": round(t_bound / n_launch * 1e3, 3),
                         "bound_shadow_aware": {"issue_cycles_per_wave_and_segment": cyc_sh, "bound_ms_per_launch": round(t_bound_sh / n_launch * 1e3, 3),
                                                "frac": round(t_bound_sh / t_priced, 4) if t_priced > 0 else None,
                                                "note": "the same mandatory slots with the ~4.5 VALU instructions per MFMA that issue for free in its shadow taken off "
                                                        "(bench/mfma_coexec.hip): the harder bound; `frac` above prices MFMA and VALU additively"},
                         "vs_fp32_valu_peak": round(achieved / FP32_VALU_PEAK_TFLOPS, 4)})
            t_loop = (seg / 64.0) * (tiles * MX_CYCLES_PER_WAVE_TILE_MEASURED) / (cus * SIMDS_PER_CU * NOMINAL_CLOCK_GHZ * 1e9)
            roof["tile_loop_measured"] = {"cycles_per_wave_tile": MX_CYCLES_PER_WAVE_TILE_MEASURED, "ms_per_launch": round(t_loop / n_launch * 1e3, 3),
                                          "source": "profiles/r03_mx_tile_rate.txt (bench/mx_tile_rate.hip: the tile loop alone, five instruction orders, 4 waves per SIMD)",
                                          "note": "what the evaluated tiles cost at the rate the tile loop reaches in isolation, at the nominal clock"}
            ph = profile_phase_split(tag) if tag else None
            if ph and depth == 1:
                broad_ms = (ph.get("broad", 0.0) + ph.get("other", 0.0)) * ms_avg
                roof["broad_phase"] = {"share_of_kernel": round(ph.get("broad", 0.0) + ph.get("other", 0.0), 4), "ms": round(broad_ms, 3),
                                       "frac_of_bound": round(t_bound / n_launch * 1e3 / broad_ms, 4) if broad_ms > 0 else None,
                                       "frac_of_measured_tile_loop": round(t_loop / n_launch * 1e3 / broad_ms, 4) if broad_ms > 0 else None,
                                       "other_phases": {k: ph[k] for k in ("narrow", "shade", "raygen", "queue") if k in ph},
                                       "source": ph["file"],
                                       "note": "the bound prices the broad phase only; this is the share of the kernel's wave-cycles the broad phase takes (tile loop + ray "
                                               "operands + top tile, -DMIRT_PHASE_TIMING build) and the bound against THAT time; the rest of the kernel is the exact narrow "
                                               "phase, shading, ray generation and the work queue, which no flop count prices"}
            note += (f"; bound = mandatory issue slots of the matrix-pipe broad phase: per wave and segment {tiles:.2f} tiles evaluated (of {tiles_all}: the tile cull skips the tiles no ray of a wave "
                     f"can touch; the count includes the top tile(s) it tests first) x (4 MFMA x 32 + 32 VALU x 2 cycles) + 6 = {cyc:.0f} SIMD-cycles, "
                     f"segments / 64 waves, {cus * SIMDS_PER_CU} SIMDs at the nominal {NOMINAL_CLOCK_GHZ} GHz; `peak` is that bound expressed in the same algorithmic flops, so frac = bound time / kernel time; "
                     "`vs_fp32_valu_peak` is the algorithmic rate against the vector-ALU peak alone, which this kernel can exceed because 8 of the 9 VALU operations per pair moved to the "
                     "matrix pipe - the metric's '% of FP32 VALU peak' is `valu_broad.frac` (render_kernel_f32, every pair on the vector ALU)")
        elif kind == 2:
            roof.update({"bound": "latency (tree walk; no brute-force flops)", "achieved": None, "peak": None, "unit": "TFLOP/s", "frac": None})
        elif tiny:
            rate = profile_valu_rate(tag) if tag else None
            roof.update({"bound": "valu issue (per-segment shading, RNG and texture gather: SURVEY.md 8d reports VALU utilisation here, not a flop fraction)",
                         "achieved": rate, "peak": 0.5, "unit": "VALU wave-instructions per SIMD-cycle", "frac": round(rate / 0.5, 4) if rate else None,
                         "achieved_source": f"profiles/{tag}_sq_pmc.txt" if rate else None,
                         "vs_fp32_valu_peak": round(achieved / FP32_VALU_PEAK_TFLOPS, 4)})
        else:
            pk = FP32_VALU_PEAK_TFLOPS if args.precision == 32 else FP64_VALU_PEAK_TFLOPS
            roof.update({"bound": "valu_fp32" if args.precision == 32 else "valu_fp64", "achieved": round(achieved, 3), "peak": pk, "unit": "TFLOP/s", "frac": round(achieved / pk, 4)})
        roof["peak_fp32_valu_nominal"] = FP32_VALU_PEAK_TFLOPS
        roof["peak_fp32_valu_measured"] = round(measured_peak.value, 1) if measured_peak.value else None
        roof["note"] = note + "; peak_fp32_valu_measured = independent v_fma_f32 chains at 8 waves/SIMD on this box (mirt_measure_fma_peak)"
        out = {
            "metric": "Mrays/s at 1080p (ray segment = one closest-hit query); % of FP32 VALU peak = valu_broad.frac (all-VALU brute-force kernel, same frame, same image)",
            "value": round(seg_all / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == 32 else "f64",
            "data": data_note(args.workload),
            "config": {
                "workload": f"{scene_name}: {scene.flat.n_spheres} spheres, {W}x{H}, {cam.pod.samples_per_pixel} spp, depth {cam.pod.max_bounce_depth}",
                "parallelism": ("one GPU: the whole frame in one launch, no exchange" if world == 1 and not use_dist else
                                f"interleaved {S}-row strips over {world} GPU(s), one process per GPU; bands ({'f32' if args.precision == 32 else 'f64'} RGB) "
                                + ("gathered to rank 0 (grouped send/recv: each peer over its own xGMI link)" if args.gather == "to_root" else "ring all-gather")
                                + f" with torch.distributed/{args.backend}, K4 on rank 0"),
                "frames_in_flight": depth,
                "seed": args.seed,
                "closest_hit": ["brute-force LDS tile loop, broad phase on the vector ALU",
                                "brute-force tile loop, the whole broad-phase test as one split-f16 contraction on the matrix pipe (two chained MFMAs per 32 rays x 32 spheres), "
                                "only for the tiles of spatially ordered spheres some ray of the wave can touch (a ball per (half) tile goes through the same contraction first; "
                                "MIRT_FLAG_ALL_PAIRS = every tile: `all_pairs`); spheres far larger than the rest (the ground) tested exactly by every ray; exact narrow phase on the "
                                "vector ALU, balanced across the wave; same (id, t) for every ray as the all-VALU kernel",
                                "bounding-ball tree (MIRT_FLAG_BVH)", "brute-force tile loop, f64",
                                "brute-force tile loop, f64: f32 broad phase on the matrix pipe, exact f64 narrow phase and shading",
                                "exact test on every sphere, no broad phase (scenes of a handful of spheres)"][kind],
            },
            "roofline": roof,
            "multi_gpu": None if per_rank is None else {
                "rccl_ranks": world, "backend": args.backend,
                "exchange": ("torch.distributed.gather of the RGB bands to rank 0 (grouped send/recv)" if args.gather == "to_root" else "all_gather_into_tensor of the RGB bands")
                            + ", then K4 (mirt_deinterleave_device) on rank 0",
                "gather": args.gather, "band_bytes_per_rank": int(rows0 * W * elem),
                "gather_bytes_per_rank": {"sent": int(rows0 * W * elem) if (args.gather == "to_root" and world > 1) or args.gather == "allgather" else 0,
                                          "received_rank0": int(world * rows0 * W * elem),
                                          "received_other_ranks": int(world * rows0 * W * elem) if args.gather == "allgather" else 0,
                                          "note": "to_root: each peer's band crosses ONE xGMI link once; allgather (ring): every rank receives all bands"},
                "band_kernel_ms_per_rank": [round(r_[0], 3) for r_ in per_rank],
                "gather_ms_per_rank": [round(r_[1], 3) for r_ in per_rank],
                "segments_per_rank": [int(r_[2]) for r_ in per_rank],
                "note": "band_kernel_ms = HIP-event time of the rank's render launch (frames in flight overlap: a launch's events span its neighbours); "
                        "gather_ms = events around the collective on the frame's stream",
            },
            "diagnostics": {
                "segments_per_step": seg_all / args.steps,
                "primary_samples_per_s": round(H * W * max(1, cam.pod.samples_per_pixel) * args.steps / elapsed, 1),
                "lane_utilisation": round(seg_all / max(iters_all * 64.0, 1.0), 4),
                "narrow_tests_per_segment": round(narrow_all / max(seg_all, 1.0), 3),
            },
        }
        if world == 1 and depth == 1 and not use_dist and not args.no_extras:
            # the same frames with 3 in flight (what the N > 1 runs do): the next frame's blocks take the CUs the last long paths leave idle
            extra = [DeviceScene(scene.flat, local_rank) for _ in range(2)]
            ring3 = [(ds, torch.cuda.Stream(), band, band8)] + [(e, torch.cuda.Stream(), torch.zeros_like(band), torch.zeros_like(band8)) for e in extra]
            o5 = cam.opts(precision=args.precision, seed=args.seed, strip_rows=S, part_index=0, part_count=1, flags=abi.FLAG_COMPACT_ROWS | accel_flags)
            n5 = max(6, args.steps)

            def run3(n):
                for f in range(n):
                    h5, q5, b5, b85 = ring3[f % 3]
                    abi.check(lib.mirt_render_device(h5.handle, C.byref(cam.pod), C.byref(o5), b5.data_ptr(), b85.data_ptr(), q5.cuda_stream, None))
                torch.cuda.synchronize()

            run3(3)
            t5 = time.perf_counter()
            run3(n5)
            dt5 = (time.perf_counter() - t5) / n5
            if not (torch.equal(ring3[0][2], ring3[1][2]) and torch.equal(ring3[0][2], ring3[2][2])):
                raise SystemExit("frames of the three ring slots differ")
            out["frames_in_flight_3"] = {"value": round(seg / args.steps / dt5 / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt5 * 1e3, 3), "frames": n5,
                                         "note": "same frame, three scene handles on three streams; the N > 1 runs keep frames in flight like this"}
            for e in extra:
                e.close()
        if world == 1 and kind in (1, 4) and mx_tiles and not args.no_extras:
            # the same frame with EVERY (ray, sphere) pair on the matrix pipe (MIRT_FLAG_ALL_PAIRS: no tile cull): round 2's kernel, the all-pairs rate
            oa = cam.opts(precision=args.precision, seed=args.seed, strip_rows=S, part_index=0, part_count=1, flags=abi.FLAG_COMPACT_ROWS | abi.FLAG_ALL_PAIRS)
            sta = abi.MirtStats()
            msa = []
            banda = torch.zeros_like(band)
            for _ in range(3):
                abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(oa), banda.data_ptr(), band8.data_ptr(), stream, C.byref(sta)))
                msa.append(sta.kernel_ms - sta.resolve_ms)
            if not torch.equal(banda, band):
                raise SystemExit("the all-pairs frame differs from the culled frame")
            mean_a = sum(msa) / len(msa)
            cyc_a = tiles_all * MX_CYCLES_PER_WAVE_TILE + MX_CYCLES_PER_WAVE_SEGMENT
            tb_a = (sta.segments / 64.0) * cyc_a / (cus * SIMDS_PER_CU * NOMINAL_CLOCK_GHZ * 1e9)
            out["all_pairs"] = {"value": round(sta.segments / (mean_a * 1e-3) / 1e6, 3), "unit": "Mrays/s", "kernel_ms": round(mean_a, 3), "kernel_ms_min": round(min(msa), 3),
                                "tiles_evaluated_per_wave_iteration": round(sta.mx_tiles / max(sta.wave_iterations, 1), 3),
                                "bound_ms_per_launch": round(tb_a * 1e3, 3), "frac": round(tb_a / (mean_a * 1e-3), 4), "same_image": True,
                                "note": "MIRT_FLAG_ALL_PAIRS: the same kernel without the tile cull - every (ray, sphere) pair through the matrix pipe, as in round 2; same image bit for bit"}
            del banda
        if world == 1 and kind == 1 and not args.no_extras:
            # the same frame with the broad phase on the vector ALU (MIRT_FLAG_VALU_BROAD): the kernel the VALU roofline describes directly
            o3 = cam.opts(precision=32, seed=args.seed, strip_rows=S, part_index=0, part_count=1, flags=abi.FLAG_COMPACT_ROWS | abi.FLAG_VALU_BROAD)
            st3 = abi.MirtStats()
            ms3 = []
            for _ in range(3):
                abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o3), band.data_ptr(), band8.data_ptr(), stream, C.byref(st3)))
                ms3.append(st3.kernel_ms - st3.resolve_ms)  # K2 alone
            fl3 = 18 * st3.static_tests + 27 * st3.moving_tests + 5 * st3.segments
            mean3 = sum(ms3) / len(ms3)
            tf3 = fl3 / (mean3 * 1e-3) / 1e12  # the mean, as for the headline; the best launch beside it
            out["valu_broad"] = {"value": round(st3.segments / (mean3 * 1e-3) / 1e6, 3), "unit": "Mrays/s", "kernel": "render_kernel_f32", "kernel_ms": round(mean3, 3),
                                 "kernel_ms_min": round(min(ms3), 3), "launches": len(ms3),
                                 "bound": "valu_fp32", "achieved": round(tf3, 3), "peak": FP32_VALU_PEAK_TFLOPS, "frac": round(tf3 / FP32_VALU_PEAK_TFLOPS, 4),
                                 "frac_best_launch": round(fl3 / (min(ms3) * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
                                 "frac_of_measured_peak": round(tf3 / measured_peak.value, 4) if measured_peak.value else None,
                                 "note": "same frame, same image, every (ray, sphere) pair on the vector ALU: the metric's % of FP32 VALU peak, in the algorithmic flops of SURVEY.md 8d "
                                         "(against the nominal 157.3 TFLOP/s and against this box's measured FMA peak)"}
        if world == 1 and args.accel == "brute" and args.precision == 32 and not args.no_extras:
            # not the metric: the same frame through the bounding-volume tree (MIRT_FLAG_BVH), bit-identical image
            o2 = cam.opts(precision=32, seed=args.seed, strip_rows=S, part_index=0, part_count=1, flags=abi.FLAG_COMPACT_ROWS | abi.FLAG_BVH)
            st2 = abi.MirtStats()
            ms2 = []
            for _ in range(3):
                abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o2), band.data_ptr(), band8.data_ptr(), stream, C.byref(st2)))
                ms2.append(st2.kernel_ms)
            out["accel_bvh"] = {"value": round(st2.segments / (min(ms2) * 1e-3) / 1e6, 3), "unit": "Mrays/s", "kernel_ms": round(min(ms2), 3),
                                "note": "same frame, closest hit through the bounding-volume tree instead of the brute-force tile loop; same image"}
        if world == 1 and args.precision == 32 and not args.no_extras:
            # not the metric: the same frame in the f64 parity mode (the reference's arithmetic operation for operation)
            band64 = torch.zeros((rows0, W, 3), dtype=torch.float64, device="cuda")
            o4 = cam.opts(precision=64, seed=args.seed, strip_rows=S, part_index=0, part_count=1, flags=abi.FLAG_COMPACT_ROWS)
            k4 = lib.mirt_closest_hit_kind(ds.handle, C.byref(cam.pod), C.byref(o4))
            st4 = abi.MirtStats()
            ms4 = []
            for _ in range(2):
                abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o4), band64.data_ptr(), band8.data_ptr(), stream, C.byref(st4)))
                ms4.append(st4.kernel_ms)
            out["f64_parity_mode"] = {"value": round(st4.segments / (min(ms4) * 1e-3) / 1e6, 3), "unit": "Mrays/s", "kernel_ms": round(min(ms4), 3),
                                      "kernel": "render_kernel_f64_mx" if k4 == 4 else "render_kernel_f64<2>",
                                      "note": "same scene and camera in binary64; its own RNG dimension layout, so a different (equally valid) image"}
            del band64
        if world == 1 and not args.no_extras:
            # report the f32-vs-f64 hit-id mismatch count of the deterministic map (camera.rs:247-254) at this size.
            # Each precision is bit-exact against its own oracle instantiation (tests/test_gpu_parity.py); the two differ at
            # silhouette pixels, where binary32 and binary64 disagree on which side of a sphere's edge a pixel corner lies.
            cam0 = build_workload(args.workload, args.width, 0, args.depth)[1]
            ids = {}
            for prec in (32, 64):
                d_id = torch.zeros((cam0.height, cam0.width), dtype=torch.int32, device="cuda")
                o0 = cam0.opts(precision=prec, seed=args.seed)
                abi.check(lib.mirt_primary_hits_device(ds.handle, C.byref(cam0.pod), C.byref(o0), d_id.data_ptr(), None, stream))
                torch.cuda.synchronize()
                ids[prec] = d_id
            mism = int((ids[32] != ids[64]).sum().item())
            out["parity"] = {"f32_vs_f64_hit_id_mismatches": mism, "pixels": cam0.width * cam0.height, "share": round(mism / (cam0.width * cam0.height), 7),
                             "note": "deterministic primary-ray hit-id maps (no-AA branch) of this workload's scene at the bench resolution, MIRT_F32 against MIRT_F64 on the GPU; "
                                     "each map is bit-exact against the oracle's instantiation of its own precision (tests/test_gpu_parity.py, tests/golden/); the f32 RADIANCE is "
                                     "held per pixel to the reference's f64 arithmetic under the same uniforms in tests/test_gpu_f32_vs_reference.py (bars in DESIGN.md 2)"}
        if world == 1 and not args.no_cpu_baseline:
            base, mt = cpu_baseline(scene, cam, args.seed, args.cpu_seconds)
            out["cpu_baseline"] = base
            out["cpu_baseline_all_cores"] = mt
            # BASELINE.json configs[0]: the reference's own default render, whole frame, CPU single thread - and the same frame on the GPU
            c1, sc1, cam1 = cpu_baseline_c1(args.seed)
            st1 = abi.MirtStats()
            ms1 = []
            o1 = cam1.opts(precision=64, seed=args.seed)
            ds1 = sc1.device_scene(local_rank)
            f1 = torch.zeros((cam1.height, cam1.width, 3), dtype=torch.float64, device="cuda")
            for _ in range(3):
                abi.check(lib.mirt_render_device(ds1.handle, C.byref(cam1.pod), C.byref(o1), f1.data_ptr(), None, stream, C.byref(st1)))
                ms1.append(st1.kernel_ms)
            c1["gpu_same_frame"] = {"precision": "f64 (parity mode)", "kernel_ms": round(min(ms1), 3), "value": round(st1.segments / (min(ms1) * 1e-3) / 1e6, 1), "unit": "Mrays/s"}
            out["cpu_baseline_c1"] = c1
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
