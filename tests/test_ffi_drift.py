"""include/mirt.h and the Rust shim ffi/rust/mirt-sys/src/lib.rs must say the same thing.  No GPU, no Rust toolchain: both files
are parsed as text (the shim has never seen a compiler - cargo / rustc do not exist in the image - so this test is the only
guard between the header and a maintainer's `cargo build`).  The twin of tests/test_abi.py's header / ctypes check.

Checked: the same set of entry points with the same number of arguments; the same MIRT_* constants (macros and enum members)
with equal values; every struct with the same field names in the same order; a compile-time size assertion in lib.rs for
every struct, equal to the size of the ctypes mirror (which tests/test_abi.py pins against the header by hand); the version
comment."""
import ctypes as C
import os
import re

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mirt.h")
SHIM = os.path.join(ROOT, "ffi", "rust", "mirt-sys", "src", "lib.rs")


def _c_source():
    src = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _rust_source():
    src = open(SHIM).read()
    return re.sub(r"//[^\n]*", "", src)


def _c_int(text):
    """A C integer expression of the kinds the header uses: 2u, 0xFFFFFFFFu, -3, (128ull << 20)."""
    t = re.sub(r"(?<=[0-9a-fA-F])(?:ull|ULL|ul|UL|u|U|l|L)\b", "", text.strip())
    assert re.fullmatch(r"[\s0-9a-fA-FxX()<>+\-*|]+", t), text
    return eval(t)  # noqa: S307 - digits and operators only, checked above


def _rust_int(text):
    t = text.strip().replace("_", "")
    assert re.fullmatch(r"[\s0-9a-fA-FxX()<>+\-*|]+", t), text
    return eval(t)  # noqa: S307


def header_constants():
    src = _c_source()
    out = {}
    for name, val in re.findall(r"^#define\s+(MIRT_[A-Z0-9_]+)\s+(.+?)\s*$", src, flags=re.M):
        if name == "MIRT_H":
            continue
        out[name] = _c_int(val)
    for body in re.findall(r"typedef enum \w+\s*\{(.*?)\}", src, flags=re.S):
        for name, val in re.findall(r"(MIRT_[A-Z0-9_]+)\s*=\s*([^,\n}]+)", body):
            out[name] = _c_int(val)
    return out


def shim_constants():
    return {n: _rust_int(v) for n, v in re.findall(r"pub const (MIRT_[A-Z0-9_]+)\s*:\s*\w+\s*=\s*([^;]+);", _rust_source())}


def _arg_count(args):
    args = args.strip()
    return 0 if args in ("", "void") else args.count(",") + 1


def header_functions():
    src = _c_source()
    return {n: _arg_count(a) for n, a in re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src)}


def shim_functions():
    return {n: _arg_count(a) for n, a in re.findall(r"pub fn (mirt_[a-z0-9_]+)\s*\(([^)]*)\)", _rust_source())}


def header_structs():
    out = {}
    for body, name in re.findall(r"typedef struct \w+\s*\{(.*?)\}\s*(\w+)\s*;", _c_source(), flags=re.S):
        fields = []
        for decl in body.split(";"):
            m = re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", decl.strip())
            if decl.strip() and m:
                fields.append(m.group(1))
        out[name] = fields
    return out


def shim_structs():
    out = {}
    for name, body in re.findall(r"pub struct (mirt_\w+)\s*\{(.*?)\n\}", _rust_source(), flags=re.S):
        out[name] = re.findall(r"(?:pub\s+)?(\w+)\s*:", body)
    return out


def test_same_entry_points_with_the_same_arity():
    h, r = header_functions(), shim_functions()
    assert sorted(h) == sorted(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert h == r, {k: (h[k], r[k]) for k in h if h[k] != r[k]}
    assert sorted(h) == sorted(abi.EXPORTS)


def test_same_constants_with_equal_values():
    h, r = header_constants(), shim_constants()
    assert sorted(h) == sorted(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert h == r, {k: (h[k], r[k]) for k in h if h[k] != r[k]}
    # and the Python mirror's copies
    assert h["MIRT_FLAG_MX_SMALL"] == abi.FLAG_MX_SMALL and h["MIRT_STATIC_SCRATCH_BYTES"] == abi.STATIC_SCRATCH_BYTES
    assert h["MIRT_ABI_VERSION"] == abi.load().mirt_abi_version()


def test_same_struct_fields_in_the_same_order():
    h, r = header_structs(), shim_structs()
    opaque = {"mirt_scene", "mirt_node"}
    assert sorted(h) == sorted(set(r) - opaque), (sorted(h), sorted(r))
    for name, fields in h.items():
        assert r[name] == fields, (name, fields, r[name])
    for name in opaque:
        assert r[name] == ["_opaque"]


def test_every_struct_has_a_size_assertion_equal_to_the_c_size():
    mirror = {"mirt_image": abi.MirtImage, "mirt_texture": abi.MirtTexture, "mirt_material": abi.MirtMaterial, "mirt_sphere": abi.MirtSphere,
              "mirt_scene_desc": abi.MirtSceneDesc, "mirt_camera": abi.MirtCamera, "mirt_opts": abi.MirtOpts, "mirt_stats": abi.MirtStats,
              "mirt_node_stats": abi.MirtNodeStats}
    asserted = {n: int(v) for n, v in re.findall(r"assert!\(std::mem::size_of::<(mirt_\w+)>\(\)\s*==\s*(\d+)\)", _rust_source())}
    assert sorted(asserted) == sorted(mirror) == sorted(header_structs())
    for name, ct in mirror.items():
        assert asserted[name] == C.sizeof(ct), (name, asserted[name], C.sizeof(ct))


def test_version_comment_and_defaults():
    head = open(SHIM).read().splitlines()[0]
    m = re.search(r"ABI version (\d+)", head)
    assert m and int(m.group(1)) == header_constants()["MIRT_ABI_VERSION"], head
    # every plain-data struct a caller fills or receives can be zero-initialised from Rust
    src = _rust_source()
    for name in ("mirt_texture", "mirt_material", "mirt_sphere", "mirt_camera", "mirt_opts", "mirt_stats", "mirt_node_stats"):
        derive = re.search(r"#\[derive\(([^)]*)\)\]\s*pub struct " + name + r"\b", src)
        assert derive and "Default" in derive.group(1), name
