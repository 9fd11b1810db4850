"""mirt_render_device_static - the entry point that allocates nothing (include/mirt.h) - against
mirt_render_device: same image bit for bit, two calls in a row, the second with a smaller input.  Needs a GPU."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, presets

pytestmark = pytest.mark.gpu


def _both(lib, ds, cam, prec, seed, flags=0):
    import torch

    real = torch.float64 if prec == abi.F64 else torch.float32
    H, W = cam.height, cam.width
    o = cam.opts(precision=prec, seed=seed, flags=flags)
    outs = []
    for fn in (lib.mirt_render_device, lib.mirt_render_device_static):
        lin = torch.full((H, W, 3), float("nan"), dtype=real, device="cuda")
        rgb8 = torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda")
        st = abi.MirtStats()
        abi.check(fn(ds.handle, C.byref(cam.pod), C.byref(o), lin.data_ptr(), rgb8.data_ptr(), None, C.byref(st)))
        outs.append((lin, rgb8, st.segments))
    return outs


@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_static_entry_point_equals_render_device_twice_in_a_row(prec):
    import torch

    lib = abi.load()
    sc = presets.bouncing_balls_scene(1)
    ds = sc.device_scene(0)
    # first call: the larger input; second: a smaller image, fewer samples, another seed - the array keeps the first
    # call's contents and the second must not see them
    for cam, seed in ((presets.bouncing_balls_camera(400, 16, 50), 1), (presets.bouncing_balls_camera(96, 3, 12), 2)):
        (a_lin, a8, a_seg), (b_lin, b8, b_seg) = _both(lib, ds, cam, prec, seed)
        assert a_seg == b_seg and a_seg > 0
        assert torch.equal(a_lin, b_lin) and not torch.isnan(b_lin).any()
        assert torch.equal(a8, b8)


def test_static_entry_point_other_front_ends_and_async():
    import torch

    lib = abi.load()
    sc = presets.bouncing_balls_scene(1)
    ds = sc.device_scene(0)
    cam = presets.bouncing_balls_camera(200, 4, 50)
    for flags in (abi.FLAG_VALU_BROAD, abi.FLAG_BVH):
        (a_lin, a8, a_seg), (b_lin, b8, b_seg) = _both(lib, ds, cam, abi.F32, 3, flags)
        assert a_seg == b_seg and torch.equal(a_lin, b_lin) and torch.equal(a8, b8)
    # stats == NULL: one memset and two kernels on the caller's stream, nothing else
    ref = _both(lib, ds, cam, abi.F32, 5)[0][0]
    out = torch.zeros_like(ref)
    s = torch.cuda.Stream()
    o = cam.opts(precision=abi.F32, seed=5)
    abi.check(lib.mirt_render_device_static(ds.handle, C.byref(cam.pod), C.byref(o), out.data_ptr(), None, s.cuda_stream, None))
    s.synchronize()
    assert torch.equal(out, ref)


def test_static_entry_point_refuses_what_does_not_fit():
    import torch

    lib = abi.load()
    sc = presets.bouncing_balls_scene(1)
    ds = sc.device_scene(0)
    big = presets.bouncing_balls_camera(1920, 64, 50)  # 771 MiB of partial sums
    o = big.opts(precision=abi.F32, seed=1)
    out = torch.zeros((big.height, big.width, 3), dtype=torch.float32, device="cuda")
    rc = lib.mirt_render_device_static(ds.handle, C.byref(big.pod), C.byref(o), out.data_ptr(), None, None, None)
    assert rc == -6 and b"static scratch" in lib.mirt_last_error()
    # the reference's default frame (400 x 225 x 100 spp) fits
    c1 = presets.bouncing_balls_camera(400, 100, 50)
    out1 = torch.zeros((c1.height, c1.width, 3), dtype=torch.float32, device="cuda")
    o1 = c1.opts(precision=abi.F32, seed=1)
    st = abi.MirtStats()
    abi.check(lib.mirt_render_device_static(ds.handle, C.byref(c1.pod), C.byref(o1), out1.data_ptr(), None, None, C.byref(st)))
    assert st.segments > 400 * 225 * 100


def test_static_entry_point_allocates_nothing_and_can_be_captured():
    """the header promises "no hipMalloc / hipFree / event or host allocation of any kind" - checked, not only
    claimed: (a) free device memory is the same before and after a first static call on a fresh handle (the occupancy cache
    and the operands were filled by mirt_scene_create); (b) the call runs under stream capture in torch's global capture
    mode, where an allocating HIP call is an error, and the replayed graph writes the frame of mirt_render_device."""
    import torch

    lib = abi.load()
    sc = presets.bouncing_balls_scene(2)
    ds = sc.device_scene(0)
    cam = presets.bouncing_balls_camera(160, 4, 20)
    o = cam.opts(precision=abi.F32, seed=8)
    out = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    abi.check(lib.mirt_render_device_static(ds.handle, C.byref(cam.pod), C.byref(o), out.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    first = out.clone()
    ref = torch.zeros_like(out)
    abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), ref.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    assert torch.equal(first, ref)
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        abi.check(lib.mirt_render_device_static(ds.handle, C.byref(cam.pod), C.byref(o), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream, None))
    torch.cuda.synchronize()
    assert not torch.equal(out, ref)  # capture records, it does not run
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
