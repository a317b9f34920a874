"""tools/rayn_dump.py: the dump format round-trips and the comparer detects differences."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "rayn_dump.py")


def _run(*a):
    return subprocess.run([sys.executable, TOOL, *a], capture_output=True, text=True)


def test_dump_and_compare(tmp_path, oracle):
    a, b, c = (str(tmp_path / n) for n in "abc")
    base = ["--scene", "s1", "--w", "32", "--h", "32", "--samples", "1", "--bounces", "2", "--tile", "1"]
    assert _run("dump", a, *base).returncode == 0
    assert _run("dump", b, *base).returncode == 0
    r = _run("compare", a, b)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr
    assert _run("dump", c, "--scene", "s2", "--w", "32", "--h", "32", "--samples", "1", "--bounces", "2", "--tile", "1").returncode == 0
    r = _run("compare", a, c)
    assert r.returncode == 1 and "DIFFERENT" in r.stdout
    assert set(os.listdir(a)) >= {"manifest.json", "samples_1d.f32", "samples_2d.f32", "scramble.f32", "fis.f32", "color.f32", "alpha.f32",
                                  "background.f32", "normal.f32", "trace.u32"}


@pytest.mark.gpu
def test_gpu_dump_equals_oracle_dump(tmp_path, oracle):
    a, b = str(tmp_path / "o"), str(tmp_path / "g")
    base = ["--scene", "s2", "--w", "48", "--h", "32", "--samples", "2", "--bounces", "3", "--tile", "2"]
    assert _run("dump", a, *base).returncode == 0
    r = _run("dump", b, *base, "--backend", "gpu")
    assert r.returncode == 0, r.stderr
    r = _run("compare", a, b)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout
