"""The alternative-reading oracle libraries (oracle/Makefile `variants`, oracle/SENSITIVITY.md): they build, render, and each is
ONE assumption read the other way - never the default.  The default oracle's arithmetic is guarded by the golden fixtures
(tests/test_oracle.py, tests/test_config_digests.py); here only the machinery the pin script relies on is checked."""
import numpy as np

from common import case, film_equal_bits


def test_variants_build_render_and_differ_where_they_should(oracle):
    oracle.build_variants()
    wd, p = case("ship", 48, 32, 2, 3)
    base_tabs = oracle.build_tables(8, 3, p.volume_marches, p.frame, 48, 32)
    ref, ctr0 = oracle.render(wd, p, base_tabs)
    differ = {}
    for v in oracle.VARIANTS:
        tabs = oracle.build_tables(8, 3, p.volume_marches, p.frame, 48, 32, variant=v)
        # the R_d tables and the scramble do not depend on any of A2..A5; the filter table does (Lerp, powers of the window)
        assert np.array_equal(tabs[0], base_tabs[0]) and np.array_equal(tabs[1], base_tabs[1]) and np.array_equal(tabs[2], base_tabs[2])
        film, ctr = oracle.render(wd, p, tabs, variant=v)
        assert ctr.paths == ctr0.paths == 48 * 32 * 8
        assert np.isfinite(film["color"]).all() and film["color"].sum() > 0
        differ[v] = not film_equal_bits(film, ref)
    # readings that touch the march / shading arithmetic change this fractal frame; the max/min readings differ only for NaN / signed zeros
    for v in ("normalize_div", "dot_plain", "normals_central", "normals_order", "lerp_alt"):
        assert differ[v], v
    assert not differ["minmax_swapped"] and not differ["minmax_ieee"]


def test_dump_tool_takes_variants_and_foreign_tables(tmp_path):
    """tools/pin_against_rayn.sh stage 2 / 3: `rayn_dump.py dump --tables-from DIR [--variant V | --fma 1]`."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "rayn_dump.py")
    args = ["--scene", "ship", "--w", "32", "--h", "32", "--samples", "1", "--bounces", "2", "--tile", "1"]
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    subprocess.run([sys.executable, tool, "dump", str(a)] + args, check=True, capture_output=True)
    subprocess.run([sys.executable, tool, "dump", str(b)] + args + ["--tables-from", str(a)], check=True, capture_output=True)
    r = subprocess.run([sys.executable, tool, "compare", str(a), str(b)], capture_output=True, text=True)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout
    subprocess.run([sys.executable, tool, "dump", str(c)] + args + ["--tables-from", str(a), "--variant", "normals_central"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, tool, "compare", str(a), str(c)], capture_output=True, text=True)
    assert r.returncode == 1 and "DIFFERENT" in r.stdout and "samples_1d" in r.stdout
