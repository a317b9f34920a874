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


# ---- the reference-side patch (bindings/rayn.patch) and the shim's use of rayn's items ------------------------------------
import shutil
import subprocess
import sys

import pytest

REFERENCE = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src")) or shutil.which("git") is None,
                                     reason="the reference checkout (build container only) and git are needed")

# std / generic-array / rand items film_hip.rs calls; everything else it calls must be defined in rayn (patched) or hip.rs
STD_METHODS = {"len", "iter", "iter_mut", "enumerate", "lock", "unwrap", "zip", "chunks_exact", "copy_from_slice", "as_ptr", "as_mut_ptr",
               "ok_or_else", "gen", "from", "seed_from_u64", "new"}


@pytest.fixture(scope="module")
def patched_src(tmp_path_factory):
    """reference src/ + bindings/rayn.patch (git apply) + hip.rs + film_hip.rs, as a maintainer would have it"""
    d = tmp_path_factory.mktemp("rayn_patched")
    shutil.copytree(os.path.join(REFERENCE, "src"), d / "src")
    subprocess.run(["git", "init", "-q", "."], cwd=d, check=True)
    patch = os.path.join(ROOT, "bindings", "rayn.patch")
    r = subprocess.run(["git", "apply", "--check", "-p1", patch], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    subprocess.run(["git", "apply", "-p1", patch], cwd=d, check=True)
    shutil.copy(os.path.join(ROOT, "bindings", "rayn_hip.rs"), d / "src" / "hip.rs")
    shutil.copy(os.path.join(ROOT, "bindings", "film_hip.rs"), d / "src" / "film_hip.rs")
    return d / "src"


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", "", text)


@needs_reference
def test_patch_applies_and_is_what_make_patch_generates(patched_src, tmp_path):
    out = tmp_path / "regen.patch"
    subprocess.run([sys.executable, os.path.join(ROOT, "bindings", "make_patch.py"), REFERENCE, str(out)], check=True, capture_output=True)
    assert out.read_text() == open(os.path.join(ROOT, "bindings", "rayn.patch")).read(), "bindings/rayn.patch is stale: run bindings/make_patch.py"
    main = (patched_src / "main.rs").read_text()
    assert "mod hip;" in main and "mod film_hip;" in main and "render_frame_into_hip" in main


@needs_reference
def test_patched_sources_keep_their_brackets_balanced(patched_src):
    """no rustc here: at least every file the patch touches (and the two shim files) must still nest (), [] and {} properly"""
    pairs = {")": "(", "]": "[", "}": "{"}
    for f in sorted(patched_src.glob("*.rs")):
        text = strip_rust_comments(f.read_text())
        text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)   # string literals
        text = re.sub(r"'(?:\\.|[^'\\])'", "' '", text)    # char literals (lifetimes have no closing quote)
        stack = []
        for ch in text:
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], f.name
        assert not stack, f.name


@needs_reference
def test_every_scene_type_of_the_closed_set_describes_itself(patched_src):
    src = {f.name: strip_rust_comments(f.read_text()) for f in patched_src.glob("*.rs")}
    # the four traits carry the defaulted method ...
    for fname, trait, ret in (("hitable.rs", "Hitable", "RaynHitable"), ("material.rs", "Material", "RaynMaterial"), ("light.rs", "Light", "RaynLight"),
                              ("camera.rs", "Camera", "RaynCamera")):
        body = re.search(r"pub trait %s\b.*?\n}\n" % trait, src[fname], re.S).group(0)
        assert re.search(r"fn describe\(&self\) -> Option<crate::hip::%s> \{\s*None\s*\}" % ret, body), trait
    # ... and every concrete type of the closed set overrides it inside its trait impl
    for fname, impl_head in (("sphere.rs", r"impl<TR: WSequenced<Wec3>> Hitable for Sphere<TR>"), ("sdf.rs", r"Hitable for TracedSDF<S>"),
                             ("material.rs", r"Material for Lambertian<AG>"), ("material.rs", r"Material for Dielectric<AG, RG>"),
                             ("material.rs", r"impl Material for Sky"), ("material.rs", r"Material for Emissive<EG>"),
                             ("light.rs", r"impl Light for SphereLight"), ("camera.rs", r"Camera for PinholeCamera<O, A, U>"),
                             ("camera.rs", r"Camera for ThinLensCamera<A, O, LA, U, F>"), ("camera.rs", r"Camera for OrthographicCamera<O, A, U>")):
        m = re.search(impl_head + r".*?\n}\n", src[fname], re.S)
        assert m and "fn describe(&self)" in m.group(0), impl_head
    assert "impl DescribeSdf for MandelBox" in src["sdf.rs"]
    # every field a describe() reads exists in its struct
    for fname, struct, fields in (("sdf.rs", "MandelBox", ["iterations", "scale_arg", "box_fold", "sphere_fold"]), ("sdf.rs", "BoxFold", ["side_length"]),
                                  ("sdf.rs", "SphereFold", ["min_radius", "fixed_radius"]), ("light.rs", "SphereLight", ["desc"]),
                                  ("camera.rs", "PinholeCamera", ["resolution", "vfov", "origin", "at", "up"]),
                                  ("camera.rs", "ThinLensCamera", ["resolution", "vfov", "aperture", "origin", "at", "up", "focus"]),
                                  ("camera.rs", "OrthographicCamera", ["resolution", "vertical_size", "origin", "at", "up"]),
                                  ("sphere.rs", "Sphere", ["transform_seq", "radius", "material"])):
        body = re.search(r"pub struct %s\b[^{]*\{(.*?)\n}" % struct, src[fname], re.S).group(1)
        for fld in fields:
            assert re.search(r"\b%s:" % fld, body), (struct, fld)


@needs_reference
def test_the_shim_only_uses_items_that_exist(patched_src):
    """Every method film_hip.rs calls, every `crate::` path and every field it reads on a rayn type is defined in the patched
    reference or in hip.rs (round 2's shim called describe() / channel_ptrs_mut that existed nowhere)."""
    src = {f.name: strip_rust_comments(f.read_text()) for f in patched_src.glob("*.rs")}
    shim = re.sub(r"#\[[^\]]*\]", "", src["film_hip.rs"])  # attributes are not calls
    everything = "\n".join(src.values())
    defined_fns = set(re.findall(r"\bfn (\w+)", everything))
    for m in set(re.findall(r"\.(\w+)\(", shim)) | set(re.findall(r"\b(\w+)\(", shim)):
        if m in STD_METHODS or m[0].isupper() or m in ("Some", "Ok", "Err", "drop", "vec", "match", "if", "for"):
            continue
        assert m in defined_fns, f"film_hip.rs calls {m}() which nothing defines"
    # `use crate::module::{Items}` and inline crate:: paths
    for mod, items in re.findall(r"use crate::(\w+)::\{?([\w, *]+)\}?;", shim):
        assert mod + ".rs" in src, mod
        for it in [i.strip() for i in items.split(",")]:
            if it != "*":
                # a definition, or a name handed to a defining macro (`srgbs! { Srgb => Vec3, f32, .. }`, src/spectrum.rs)
                assert re.search(r"\b(struct|enum|trait|type|fn|const) %s\b|\b%s =>" % (it, it), src[mod + ".rs"]), (mod, it)
    for mod, item in re.findall(r"crate::(\w+)::([A-Z_][A-Z0-9_]+)\b", shim):
        assert re.search(r"pub(\(crate\))? const %s\b" % item, src[mod + ".rs"]), (mod, item)
    # fields read through self / world / integrator / sample_sets / fis must be visible from another module
    visible = {("film.rs", "res"), ("film.rs", "channels"), ("film.rs", "progressive_epoch"), ("filter.rs", "inverse_cdf"), ("sampler.rs", "samples_1d"),
               ("sampler.rs", "samples_2d"), ("integrator.rs", "max_bounces"), ("integrator.rs", "volume_marches"), ("world.rs", "hitables"),
               ("world.rs", "materials"), ("world.rs", "lights"), ("world.rs", "cameras"), ("world.rs", "volume_params"),
               ("volume.rs", "coeff_scattering"), ("volume.rs", "coeff_extinction")}
    for fname, fld in visible:
        assert re.search(r"\.%s\b" % fld, shim), fld
        assert re.search(r"pub(\(crate\))? %s:" % fld, src[fname]), (fname, fld)
    # constants and helpers the describe() overrides take from hip.rs
    hip = src["hip.rs"]
    for name in set(re.findall(r"crate::hip::(\w+)", everything)):
        assert re.search(r"pub (const|struct|enum|fn) %s\b" % name, hip), name
    for setter in set(re.findall(r"\bd\.(set_\w+)\(", src["camera.rs"])):
        assert "pub fn %s(" % setter in hip, setter


@needs_reference
def test_dump_hooks_only_use_items_that_exist(patched_src):
    """RAYN_DUMP=<dir>[,<tile>] (N2; the hooks live INSIDE bindings/rayn.patch since round 4): src/dump.rs is created by the patch,
    and every function, method, field, enum variant and constant the hooks in src/film.rs touch is defined by the patched reference -
    the check a compiler would make first (r3's prose snippet called a `bin_lens()` that existed nowhere)."""
    src = {f.name: strip_rust_comments(f.read_text()) for f in patched_src.glob("*.rs")}
    assert "dump.rs" in src and re.search(r"^mod dump;", src["main.rs"], re.M)
    film, dump = src["film.rs"], src["dump.rs"]
    # every crate::dump:: function the film calls is a pub fn of dump.rs with that many parameters
    for fn, args in re.findall(r"crate::dump::(\w+)\(((?:[^()]|\([^()]*\))*)\)", film):
        m = re.search(r"pub fn %s\((.*?)\)\s*(->|\{)" % fn, dump, re.S)
        assert m, fn
        n_params = len([p for p in m.group(1).split(",") if p.strip()])
        n_args = len([a for a in re.split(r",(?![^()]*\))", args) if a.strip()])
        assert n_params == n_args, (fn, n_params, n_args)
    assert {"config", "write_f32", "write_u32", "write_manifest"} <= set(re.findall(r"crate::dump::(\w+)\(", film))
    # HitStore::bin_lens exists and iterates the bins the trace loop walks
    assert re.search(r"pub fn bin_lens<'a>\(&'a self\) -> impl Iterator<Item = usize> \+ 'a", src["hitable.rs"])
    assert "hit_store.bin_lens().enumerate()" in film
    # fields the hooks read
    tile_struct = re.search(r"pub struct Tile<[^{]*\{(.*?)\n}", film, re.S).group(1)
    assert re.search(r"\b_index: usize", tile_struct) and "tile._index" in film
    ray_macro = src["ray.rs"]
    for fld in ("tile_coord", "sample", "valid"):
        assert re.search(r"pub %s:" % fld, ray_macro) and ("wsp.ray.%s[lane]" % fld) in film, fld
    assert re.search(r"pub ray: WRay", src["hitable.rs"])  # WShadingPoint::ray
    assert re.search(r"pub samples_1d: Vec<f32>", src["sampler.rs"]) and re.search(r"pub samples_2d: Vec<f32>", src["sampler.rs"])
    assert re.search(r"pub\(crate\) inverse_cdf:", src["filter.rs"])
    for const in ("MAX_INDIRECT_BOUNCES", "VOLUME_MARCHES_PER_SAMPLE"):
        assert re.search(r"pub const %s: usize" % const, src["setup.rs"]), const
    # the four channel kinds dump_channels matches are exactly the ones declare_channels! declares (a missing arm would not compile)
    declared = re.findall(r"^\s{4}(\w+) => \{\s*storage:", film, re.M)
    matched = re.findall(r"ChannelStorage::(\w+)\(buf\) =>", re.search(r"fn dump_channels.*?\n    }\n", film, re.S).group(0))
    assert sorted(declared) == sorted(matched) == ["Alpha", "Background", "Color", "WorldNormal"]
    # the closure handed to integrate_tiles must stay Copy + Send + Sync (src/film.rs:630-633): it may capture the dump
    # configuration only by shared reference
    assert "let dump_ref = dump_cfg.as_ref();" in film and "dump_cfg" not in re.search(r"integrate_tiles\(tiles, samples \* 4, \|tile\| \{(.*?)\n        \}\);", film, re.S).group(1)


def test_a_rayn_side_dump_is_accepted_by_the_comparer(tmp_path):
    """The manifest src/dump.rs writes (format string in bindings/make_patch.py) carries every key `rayn_dump.py compare` checks, and a
    directory in that layout compares IDENTICAL with an oracle dump of the same parameters: the one-command pin of tools/pin_against_rayn.sh
    ends in exactly this comparison."""
    import json
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "rayn_dump.py")
    a = tmp_path / "oracle"
    subprocess.run([sys.executable, tool, "dump", str(a), "--scene", "s2", "--w", "32", "--h", "32", "--samples", "1", "--bounces", "2", "--tile", "1"], check=True, capture_output=True)
    b = tmp_path / "rayn_like"
    shutil.copytree(a, b)
    fmt = re.search(r'"\{\{(.*?)\}\}\\\\n",', open(os.path.join(ROOT, "bindings", "make_patch.py")).read(), re.S).group(1)
    keys = re.findall(r'\\\\"(\w+)\\\\":', fmt)
    assert {"width", "height", "spp", "max_bounces", "volume_marches", "frame", "trace_tile", "tile", "time_range"} <= set(keys)
    m = json.load(open(a / "manifest.json"))
    rayn_manifest = {"scene": "rayn setup::setup()", "width": m["width"], "height": m["height"], "SAMPLES": m["SAMPLES"], "spp": m["spp"], "max_bounces": m["max_bounces"],
                     "volume_marches": m["volume_marches"], "frame": m["frame"], "time_range": m["time_range"], "tile": m["tile"], "trace_tile": 1, "backend": "rayn"}
    assert set(rayn_manifest) == set(keys)
    json.dump(rayn_manifest, open(b / "manifest.json", "w"))
    r = subprocess.run([sys.executable, tool, "compare", str(a), str(b)], capture_output=True, text=True)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr


@needs_reference
def test_dump_only_patch_applies_alone_and_is_current(tmp_path):
    """bindings/rayn_dump.patch = ONLY the RAYN_DUMP hooks (what tools/pin_against_rayn.sh applies: pinning the CPU oracle needs no GPU
    library): it applies to a pristine copy of the reference on its own, leaves `mod hip` / `describe()` out, and is what
    bindings/make_patch.py regenerates."""
    d = tmp_path / "rayn"
    shutil.copytree(os.path.join(REFERENCE, "src"), d / "src")
    subprocess.run(["git", "init", "-q", "."], cwd=d, check=True)
    patch = os.path.join(ROOT, "bindings", "rayn_dump.patch")
    r = subprocess.run(["git", "apply", "--check", "-p1", patch], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    subprocess.run(["git", "apply", "-p1", patch], cwd=d, check=True)
    main = (d / "src" / "main.rs").read_text()
    assert "mod dump;" in main and "mod hip;" not in main and "RAYN_HIP" not in main
    assert (d / "src" / "dump.rs").exists() and "fn describe" not in (d / "src" / "hitable.rs").read_text()
    assert "bin_lens" in (d / "src" / "hitable.rs").read_text() and "crate::dump::write_u32" in (d / "src" / "film.rs").read_text()
    out = tmp_path / "regen.patch"
    subprocess.run([sys.executable, os.path.join(ROOT, "bindings", "make_patch.py"), REFERENCE, str(out)], check=True, capture_output=True)
    assert (tmp_path / "regen_dump.patch").read_text() == open(patch).read(), "bindings/rayn_dump.patch is stale: run bindings/make_patch.py"
    # the one-command pin script exists, is executable and refers to this patch and to the comparer
    sh = os.path.join(ROOT, "tools", "pin_against_rayn.sh")
    text = open(sh).read()
    assert os.access(sh, os.X_OK) and "rayn_dump.patch" in text and "rayn_dump.py" in text and "cargo" in text
