"""N1: the Rust binding in bindings/rayn_hip.rs cannot be compiled here (no Rust toolchain), but it
must not drift from the ABI.  Its #[repr(C)] field lists are parsed and compared - names, order, scalar types - with the C
structs of include/rayn_hip.h and with the ctypes mirror; the struct sizes computed from the Rust field lists (C layout
rules) must equal what the compiled library reports through rayn_hip_sizeof; every extern fn must be an exported symbol."""
import ctypes as C
import os
import re

from rayn_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = open(os.path.join(ROOT, "bindings", "rayn_hip.rs")).read()
HDR = open(os.path.join(ROOT, "include", "rayn_hip.h")).read()

RUST_SCALARS = {"u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8), "f64": (8, 8)}
PAIRS = [("RaynVec3", "rayn_vec3", _abi.Vec3, None), ("RaynHitable", "rayn_hitable", _abi.Hitable, 3), ("RaynMaterial", "rayn_material", _abi.Material, 4),
         ("RaynLight", "rayn_light", _abi.Light, 5), ("RaynCamera", "rayn_camera", _abi.Camera, 6), ("RaynWorldDesc", "rayn_world_desc", _abi.WorldDesc, 0),
         ("RaynFrameParams", "rayn_frame_params", _abi.FrameParams, 1), ("RaynStats", "rayn_stats", _abi.Stats, 2)]


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\n\}", RS, re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        out[m.group(1)] = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+):\s*([^,]+),", body)]
    return out


def c_structs():
    out = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(\w+);", HDR, re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, names = decl.split(None, 1)
            for n in names.split(","):
                n = n.strip()
                arr = re.match(r"(\w+)\[(\w+)\]", n)
                fields.append((arr.group(1), f"[{ty}; {arr.group(2)}]") if arr else (n, ty))
        out[m.group(2)] = fields
    return out


def layout(fields, structs):
    """(size, align) of a Rust #[repr(C)] struct from its field list."""
    off, align = 0, 1
    for _, ty in fields:
        arr = re.match(r"\[(\w+);\s*(\w+)\]", ty)
        n = 1
        if arr:
            ty, n = arr.group(1), int(arr.group(2))
        s, a = RUST_SCALARS[ty] if ty in RUST_SCALARS else layout(structs[ty], structs)
        off = (off + a - 1) // a * a + s * n
        align = max(align, a)
    return (off + align - 1) // align * align, align


C2RUST = {"uint32_t": "u32", "float": "f32", "uint64_t": "u64", "double": "f64", "rayn_vec3": "RaynVec3", "rayn_hitable": "RaynHitable",
          "rayn_material": "RaynMaterial", "rayn_light": "RaynLight", "rayn_camera": "RaynCamera"}
CONSTS = {"RAYN_MAX_HITABLES": "16", "RAYN_MAX_MATERIALS": "16", "RAYN_MAX_LIGHTS": "16"}


def test_repr_c_fields_match_the_header_and_ctypes():
    rs, cs = rust_structs(), c_structs()
    assert len(rs) == len(PAIRS)
    for rname, cname, ctype, _ in PAIRS:
        want = []
        for n, ty in cs[cname]:
            arr = re.match(r"\[(\w+); (\w+)\]", ty)
            want.append((n, f"[{C2RUST[arr.group(1)]}; {CONSTS[arr.group(2)]}]" if arr else C2RUST[ty]))
        assert rs[rname] == want, rname
        assert [n for n, _ in rs[rname]] == [f[0] for f in ctype._fields_], rname
        assert layout(rs[rname], rs)[0] == C.sizeof(ctype), rname


def test_sizes_match_the_compiled_library():
    _lib.build()
    L = _lib.lib()
    rs = rust_structs()
    for rname, _, _, which in PAIRS:
        if which is not None:
            assert layout(rs[rname], rs)[0] == L.rayn_hip_sizeof(which), rname


def test_extern_fns_are_exported_and_declared():
    fns = re.findall(r"pub fn (rayn_\w+)\(", RS)
    assert len(fns) >= 8 and "rayn_hip_create_multi" in fns
    declared = set(re.findall(r"\b(rayn_[a-z0-9_]+)\s*\(", HDR))
    L = _lib.lib()
    for f in fns:
        assert f in declared and hasattr(L, f), f
    # the film-side shim only calls what the binding declares
    shim = open(os.path.join(ROOT, "bindings", "film_hip.rs")).read()
    for f in set(re.findall(r"\b(rayn_hip_\w+)\(", shim)):
        assert f in fns, f
