"""The C-ABI library loads without a GPU and exports exactly what include/mirt.h declares.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mirt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = abi.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.mirt_abi_version() == 3


def test_struct_layouts_match_the_header():
    # sizes computed by hand from include/mirt.h (natural alignment, 8-byte pointers)
    assert C.sizeof(abi.MirtImage) == 16
    assert C.sizeof(abi.MirtTexture) == 8 + 24 + 24 + 8
    assert C.sizeof(abi.MirtMaterial) == 8 + 8 + 24 + 8 + 8 + 8
    assert C.sizeof(abi.MirtSphere) == 24 + 24 + 8 + 8
    assert C.sizeof(abi.MirtSceneDesc) == 16 + 32
    assert C.sizeof(abi.MirtCamera) == 7 * 24 + 8 + 16
    assert C.sizeof(abi.MirtOpts) == 48
    assert C.sizeof(abi.MirtStats) == 88


def test_host_only_helpers():
    lib = abi.load()
    # Philox block computed by the library's host path must be the published function
    c = (C.c_uint32 * 4)(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344)
    k = (C.c_uint32 * 2)(0xA4093822, 0x299F31D0)
    o = (C.c_uint32 * 4)()
    lib.mirt_philox4x32_10(c, k, o)
    assert list(o) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    # interleaved strips: 1080 rows, strips of 8, 8 parts -> 135 strips; parts 0..6 own 17 strips, part 7 owns 16
    rows = [lib.mirt_part_rows(1080, 8, i, 8) for i in range(8)]
    assert rows == [136] * 7 + [128] and sum(rows) == 1080
    assert lib.mirt_part_rows(225, 8, 0, 1) == 225
    assert [lib.mirt_part_rows(225, 8, i, 2) for i in range(2)] == [8 * 14 + 1, 8 * 14]
    assert lib.mirt_part_rows(5, 0, 0, 3) == 5 and lib.mirt_part_rows(5, 0, 1, 3) == 0


def test_errors_do_not_abort():
    lib = abi.load()
    h = C.c_void_p()
    assert lib.mirt_scene_create(None, 0, C.byref(h)) == -1
    assert b"null" in lib.mirt_last_error()
    d = abi.MirtSceneDesc()
    d.n_spheres = 3  # count without an array
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    s = (abi.MirtSphere * 1)()
    s[0].material = 5
    d = abi.MirtSceneDesc(1, 0, 0, 0, s, None, None, None)
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1 and b"material" in lib.mirt_last_error()
    lib.mirt_scene_destroy(None)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(abi, "_lib", None)
    monkeypatch.setattr(abi, "LIB_PATH", "/nonexistent/libmirt.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        abi.load()
