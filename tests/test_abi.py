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
