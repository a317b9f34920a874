"""The C++ host mirror (include/rayn_host.hpp): flattening of setup::setup() equals the Python mirror's
byte for byte (CPU), and a render through it equals the oracle (GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import film_equal_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_mirror")


@pytest.fixture(scope="module")
def exe():
    from rayn_amd import _lib
    _lib.build()
    csrc = os.path.join(ROOT, "rayn_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "host_mirror.cpp"), "-o", EXE,
                           f"-L{csrc}", "-lrayn_hip", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


@pytest.mark.parametrize("volumes", [0, 1])
def test_world_desc_matches_python_mirror(exe, tmp_path, volumes):
    from rayn_amd import _abi, setup as S
    out = str(tmp_path / "desc.bin")
    subprocess.check_call([exe, "desc", "1280", "720", str(volumes), out])
    raw = open(out, "rb").read()
    cam, world = S.setup((1280, 720), volumes=bool(volumes))
    mine = bytes(memoryview(world.to_desc(cam)).cast("B"))
    assert len(raw) == C.sizeof(_abi.WorldDesc) == len(mine)
    assert raw == mine


@pytest.mark.gpu
@pytest.mark.parametrize("n_devices", [0, 2])
def test_cpp_film_render_matches_oracle(exe, tmp_path, oracle, n_devices):
    """A compiled C++ host through the C ABI: single-device Film and a multi-device Film (two entries on GPU 0)."""
    from rayn_amd import params as P, setup as S
    W, H, samples, bounces = 48, 32, 2, 3
    out = str(tmp_path / "film.bin")
    subprocess.check_call([exe, "render", str(W), str(H), str(samples), str(bounces), "1", out, str(n_devices)])
    raw = np.fromfile(out, np.float32)
    n = W * H
    got = {"color": raw[:3 * n].reshape(H, W, 3), "alpha": raw[3 * n:4 * n].reshape(H, W),
           "background": raw[4 * n:7 * n].reshape(H, W, 3), "normal": raw[7 * n:10 * n].reshape(H, W, 3)}
    cam, world = S.setup((W, H), volumes=True)
    p = P.frame_params(W, H, samples, bounces)
    ref, _ = oracle.render(world.to_desc(cam), p, oracle.build_tables(4 * samples, bounces, 2, 1, W, H))
    assert film_equal_bits(got, ref)
