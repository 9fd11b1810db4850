"""mirt_node_* - one host thread, N GPUs inside libmirt.so, RCCL exchange - on the GPUs this box has (the
round's box has ONE: RCCL then runs with a communicator of one rank; N = 2..8 ranks are rehearsed on that one GPU through
tests/stub_rccl, see the last test).  Needs a GPU."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, presets
from raytracer_amd.api import NodeScene

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


@pytest.fixture(scope="module")
def node_and_scene():
    sc = presets.bouncing_balls_scene(1)
    n_gpus = min(abi.load().mirt_device_count(), 8)
    node = NodeScene(sc.flat, list(range(n_gpus)))
    yield node, sc, n_gpus
    node.close()


@pytest.mark.parametrize("gather", [abi.GATHER_TO_ROOT, abi.GATHER_ALLGATHER])
@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_node_render_equals_single_gpu_render(node_and_scene, gather, prec):
    node, sc, n = node_and_scene
    cam = presets.bouncing_balls_camera(200, 6, 50)
    ref, ref8 = cam.render_arrays(sc, precision=prec, seed=4)
    seg = cam.last_stats["segments"]
    lin, rgb8 = cam.render_arrays_node(node, precision=prec, seed=4, gather=gather)
    st = cam.last_stats
    assert st["n_devices"] == n and st["gather"] == gather and st["segments"] == seg == sum(st["band_segments"])
    # n = the GPUs this box has: on a one-GPU box this pass says nothing about N > 1 against real RCCL (only the stub covers that)
    assert np.array_equal(bits(lin), bits(ref)) and np.array_equal(rgb8, ref8), f"node frame differs from the one-GPU frame (n_gpus = {n}, real RCCL with {n} rank(s))"
    assert st["gather_ms"] > 0 and all(ms > 0 for ms in st["band_kernel_ms"]), f"n_gpus = {n}"
    # bytes only: the bands travel as u8
    _, only8 = cam.render_arrays_node(node, precision=prec, seed=4, gather=gather, want_linear=False)
    assert np.array_equal(only8, ref8)


def test_node_render_odd_sizes_and_front_ends(node_and_scene):
    node, sc, _ = node_and_scene
    for w, spp, strip, accel in ((97, 3, 3, "bvh"), (33, 0, 1, "valu"), (250, 2, 16, None)):
        cam = presets.bouncing_balls_camera(w, spp, 12)
        ref, ref8 = cam.render_arrays(sc, precision=abi.F32, seed=2, accel=accel)
        lin, rgb8 = cam.render_arrays_node(node, precision=abi.F32, seed=2, accel=accel, strip_rows=strip, gather=abi.GATHER_TO_ROOT)
        assert np.array_equal(bits(lin), bits(ref)) and np.array_equal(rgb8, ref8)


def test_node_argument_errors(node_and_scene):
    node, sc, _ = node_and_scene
    lib = abi.load()
    cam = presets.bouncing_balls_camera(64, 1, 5)
    out = np.zeros((cam.height, cam.width, 3), np.uint8)
    o = cam.opts(precision=abi.F32, part_index=0, part_count=2)
    assert lib.mirt_node_render(node.handle, C.byref(cam.pod), C.byref(o), abi.GATHER_TO_ROOT, None, out.ctypes.data, None) == -1
    o = cam.opts(precision=abi.F32, part_count=0)
    assert lib.mirt_node_render(node.handle, C.byref(cam.pod), C.byref(o), 7, None, out.ctypes.data, None) == -1
    assert lib.mirt_node_render(node.handle, C.byref(cam.pod), C.byref(o), abi.GATHER_TO_ROOT, None, None, None) == -1
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.mirt_node_create(C.byref(sc.flat.desc), ids, 2, 0, C.byref(h)) == -1 and b"distinct" in lib.mirt_last_error()
    ids = (C.c_int * 1)(99)
    assert lib.mirt_node_create(C.byref(sc.flat.desc), ids, 1, 0, C.byref(h)) == -2


def test_node_n_ranks_on_one_gpu_through_the_stub():
    """N = 2, 3, 5, 8 ranks of mirt_node on ONE GPU: MIRT_NODE_SHARED_DEVICES plus tests/stub_rccl (HIP copies behind RCCL's
    entry points).  Covers what a one-GPU box otherwise cannot: the exchange slots and offsets, the send-to-root and
    all-gather call patterns, the scatter of N bands and the per-rank statistics.  Runs in a child process without torch
    (torch holds the real librccl.so.1)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = os.path.join(root, "tests", "stub_rccl")
    assert os.path.exists(os.path.join(stub, "librccl.so.1")), "tests/stub_rccl/librccl.so.1 missing: run __graft_entry__.build()"
    env = dict(os.environ, PYTHONPATH=root, LD_LIBRARY_PATH=stub + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([sys.executable, "-u", os.path.join(stub, "node_ranks.py")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "STUB_NODE_OK 32" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_node_stays_usable_after_a_failed_render(node_and_scene):
    """A failed node render must leave no replica `pending` (every failure after the first launch now goes through
    one way out that collects and drains all launched replicas).  From outside the library a failure can only be provoked in
    a replica's own launch - an f64 bounce limit no attenuation stack holds - so with one GPU this pins the contract (error
    code, node still renders the right frame afterwards) rather than the multi-replica unwinding itself."""
    node, sc, n = node_and_scene
    lib = abi.load()
    cam = presets.bouncing_balls_camera(64, 2, 5)
    ref, ref8 = cam.render_arrays(sc, precision=abi.F32, seed=3)
    bad = presets.bouncing_balls_camera(64, 2, 5)
    bad.pod.max_bounce_depth = 0xFFFFFFF0  # 24 B x depth x 256 lanes per block: beyond the 8 GiB limit for a single block
    out = np.zeros((cam.height, cam.width, 3), np.uint8)
    o = bad.opts(precision=abi.F64, part_count=0)
    rc = lib.mirt_node_render(node.handle, C.byref(bad.pod), C.byref(o), abi.GATHER_TO_ROOT, None, out.ctypes.data, None)
    assert rc == -6, (rc, lib.mirt_last_error())
    lin, rgb8 = cam.render_arrays_node(node, precision=abi.F32, seed=3)
    assert np.array_equal(bits(lin), bits(ref)) and np.array_equal(rgb8, ref8), f"n_gpus = {n}"
