"""The C-ABI library loads without a GPU, exports every symbol include/rayn_hip.h declares, and its
struct layouts match the ctypes mirror.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from rayn_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    hdr = open(os.path.join(ROOT, "include", "rayn_hip.h")).read()
    declared = set(re.findall(r"\b(rayn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts(L):
    for i, t in enumerate([_abi.WorldDesc, _abi.FrameParams, _abi.Stats, _abi.Hitable, _abi.Material, _abi.Light, _abi.Camera]):
        assert C.sizeof(t) == L.rayn_hip_sizeof(i), t.__name__


def test_fma_policy_matches_oracle(L, oracle):
    assert L.rayn_hip_fma_policy() == 0
    assert oracle.lib().oracle_fma_policy() == 0
    assert oracle.lib(fma=True).oracle_fma_policy() == 1


def test_tile_count_quirk(L):
    # (res + res % tile) / tile, src/film.rs:399-404: widths 17..23 with 16-px tiles give ONE column
    assert L.rayn_tile_count(1920, 1080, 16, 16) == 120 * 68
    assert L.rayn_tile_count(20, 16, 16, 16) == 1
    assert L.rayn_tile_count(24, 16, 16, 16) == 2
    assert L.rayn_tile_count(256, 256, 16, 16) == 256


def test_create_without_gpu_is_an_error_not_a_fallback(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert L.rayn_hip_create(0, C.byref(h)) != 0
    import rayn_amd
    with pytest.raises(rayn_amd.film.RaynHipError):
        rayn_amd.Context(0)


def test_bench_without_gpu_exits_loudly():
    """bench.py has no CPU path either: without a GPU it stops with a message instead of timing the oracle."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())  # and prints no result line


def test_traced_sdf_transform_maps_to_the_hitable_fields():
    """EXTENSION plumbing (host logic, no GPU): TracedSDF.transform_seq -> rayn_hitable.center / animated / center_vel,
    and the default keeps the reference's zero, non-animated origin."""
    import rayn_amd as R
    from rayn_amd import setup as S
    cam, world = S.setup_s1((32, 32))
    d = world.to_desc(cam)
    sdf = [h for h in d.hitables[: d.n_hitables] if h.kind == R._abi.HITABLE_TRACED_SDF]
    assert len(sdf) == 1 and sdf[0].animated == 0 and (sdf[0].center.x, sdf[0].center.y, sdf[0].center.z) == (0.0, 0.0, 0.0)
    cam, world = S.setup_s3((32, 32))
    d = world.to_desc(cam)
    sdf = [h for h in d.hitables[: d.n_hitables] if h.kind == R._abi.HITABLE_TRACED_SDF][0]
    assert sdf.animated == 1 and abs(sdf.center_vel.x + 0.6) < 1e-6 and d.camera.animated & 1
    world.hitables[1].transform_seq = R.vec3(0.5, 0.25, -1.0)
    sdf = [h for h in world.to_desc(cam).hitables[:8] if h.kind == R._abi.HITABLE_TRACED_SDF][0]
    assert sdf.animated == 0 and (sdf.center.x, sdf.center.y, sdf.center.z) == (0.5, 0.25, -1.0)


def test_product_source_carries_no_experiment_switches():
    """The product kernels hold exactly one code path per policy: no wrong-by-design ablation builds, no rejected layouts of the
    fold block, no opt-in kernels.  Measured variants live in git history (tools/variants/README.md)."""
    banned = re.compile(r"ABLATE|RAYN_FOLD_(BRANCHFREE|DENSE_BELOW|LIKELY|PLAIN_IF)|RAYN_PACKED_FOLD|RAYN_COUNT_(FOLDS|TRIPS)|"
                        r"shadow_scan|setup_stride|SETUP_STRIDE|RAYN_FAST_DETMATH|\bSCAN\b")
    csrc = os.path.join(ROOT, "rayn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            for n, line in enumerate(open(os.path.join(csrc, f)), 1):
                assert not banned.search(line), f"{f}:{n}: {line.strip()}"


def test_variant_builds_are_never_loaded_silently(L, monkeypatch):
    assert _lib.build_variant() == ""  # the in-tree library is the product build
    monkeypatch.setenv("RAYN_HIP_LIB", "/nonexistent/librayn_hip_exp.so")
    monkeypatch.delenv("RAYN_HIP_ALLOW_VARIANT", raising=False)
    with pytest.raises(_lib.RaynHipError, match="RAYN_HIP_ALLOW_VARIANT"):
        _lib._lib_path()
    monkeypatch.setenv("RAYN_HIP_ALLOW_VARIANT", "1")
    assert _lib._lib_path() == "/nonexistent/librayn_hip_exp.so"
