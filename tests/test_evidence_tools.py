"""The evidence tooling (tools/pmc_join.py, tools/pmc_to_json.py) and the staleness stamp bench.py relies on: kernel names with
template commas, the gfx950 byte corrections, the VALU issue figures, and that the committed PMC files carry the hash of the
kernel sources in the tree (otherwise bench.py - correctly - refuses to quote them; the check then SKIPS with a reminder)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    r = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_pmc_join_and_json(tmp_path):
    d = tmp_path
    (d / "gpurun_out").mkdir()
    (d / "a.csv").write_text("kernel,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_THREAD_CYCLES_VALU,SQ_BUSY_CYCLES\n"
                             "void rayn_p0::k_shadow1,4096000,4000000,192000000,1\nvoid foo<a, b>,1,2,3,4\n")
    (d / "b.csv").write_text("kernel,GRBM_GUI_ACTIVE,SQ_WAVES\nvoid rayn_p0::k_shadow1,80000,7\n")
    joined = _run([os.path.join(ROOT, "tools", "pmc_join.py"), str(d / "a.csv"), str(d / "b.csv")], str(d))
    rows = {ln.rsplit(",", 7)[0]: ln.rsplit(",", 7)[1:] for ln in joined.strip().splitlines()[1:]}
    assert rows["void rayn_p0::k_shadow1"][-1] == "0.4000"  # 4096000 / (80000 / 8 * 1024)
    assert rows["void foo<a, b>"][-1] == "0.0000"           # no cycle count: no rate
    (d / "sq.csv").write_text(joined)
    (d / "gpurun_out" / "pmc_fetch_t.csv").write_text("kernel,FETCH_SIZE\nvoid rayn_p0::k_shadow1,1000\n")
    (d / "gpurun_out" / "pmc_write_t.csv").write_text("kernel,WRITE_SIZE\nvoid rayn_p0::k_shadow1,500\n")
    (d / "calls.csv").write_text('"Name","Calls","TotalDurationNs"\n"void rayn_p0::k_shadow1<false, false>(int)",4,2000000\n')
    _run([os.path.join(ROOT, "tools", "pmc_to_json.py"), "t", str(d / "out.json"), str(d / "calls.csv"), str(d / "sq.csv")], str(d))
    j = json.load(open(d / "out.json"))
    k = j["kernels"]["k_shadow1"]
    assert k["fetch_bytes"] == 1000 * 1024 * 2 and k["write_bytes"] == 500 * 1024  # KiB; FETCH_SIZE x2 on gfx950
    assert k["launches"] == 4 and k["hbm_bytes_per_launch"] == (2048000 + 512000) / 4
    assert abs(k["valu_inst_per_cycle_simd"] - 0.4) < 1e-12 and abs(k["lanes_enabled"] - 0.75) < 1e-12
    sys.path.insert(0, ROOT)
    from bench import kernel_source_hash
    assert j["source_hash"] == kernel_source_hash()


def test_committed_pmc_files_match_the_kernel_sources():
    sys.path.insert(0, ROOT)
    from bench import kernel_source_hash
    for wl in ("c3", "c2", "bulb3"):
        path = os.path.join(ROOT, "profiles", f"r05_pmc_hbm_{wl}.json")
        if not os.path.exists(path):
            pytest.skip(f"{path} not collected yet (tools/gpu_round.sh)")
        j = json.load(open(path))
        if j["source_hash"] != kernel_source_hash():  # legitimate while kernels are being changed: bench.py then quotes no traffic
            pytest.skip(f"profiles/r05_pmc_hbm_{wl}.json was measured on other kernel sources: re-run tools/passes_r05/gpu_round_r05.sh before the round ends")
        dom = j["kernels"]["k_extend1" if wl == "c2" else "k_shadow1"]
        assert dom["hbm_bytes_per_launch"] > 0 and 0.2 < dom["valu_inst_per_cycle_simd"] <= 0.5 and 0.5 < dom["lanes_enabled"] <= 1.0


def test_usable_cpus_respects_the_cgroup_quota(monkeypatch, tmp_path):
    """bench.py's CPU leg runs on the CPUs the container may USE (the GPU boxes show 256 threads under a 16-CPU quota)."""
    sys.path.insert(0, ROOT)
    import builtins
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("200000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert bench.usable_cpus() == min(2, os.cpu_count() or 1)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference checkout exists in the build container only")
@pytest.mark.parametrize("reading,rc,verdict", [
    ("same", 0, "PINNED: the oracle reproduces rayn bit for bit"),
    ("tables", 0, "PINNED up to the host tables"),
    ("normals_order", 1, "IDENTICAL under: normals_order"),
    ("fma", 1, "IDENTICAL under: fma"),
])
def test_pin_script_dry_run_reaches_every_outcome(reading, rc, verdict):
    """tools/pin_against_rayn.sh end to end without a Rust toolchain (MOCK_RAYN): copy of the REAL checkout, bindings/rayn_dump.patch applied,
    the three constant edits verified, then the oracle plays rayn under a known reading in rayn's own dump layout (src/dump.rs manifest keys) -
    and the script must conclude exactly that: pinned at stage 1, pinned at stage 2 (only the host tables differ), or stage 3 naming the reading
    that is IDENTICAL.  The day a toolchain exists the script cannot fail for script reasons.  (Small frame here; the same run at rayn's
    shipped 1280x720x8spp is logged in profiles/r05_pin_mock_shipped.txt.)"""
    env = dict(os.environ, MOCK_RAYN=reading, PYTHON=sys.executable)
    r = subprocess.run(["sh", os.path.join(ROOT, "tools", "pin_against_rayn.sh"), "/root/reference", "48", "32", "1", "2"], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == rc, r.stdout[-1500:] + r.stderr[-1500:]
    assert verdict in r.stdout, r.stdout[-1500:]
    if rc == 1:  # stage 3 went through every reading: exactly one is identical
        assert r.stdout.count("\nIDENTICAL") == 1 and "== stage 3" in r.stdout


def test_pin_script_refuses_a_checkout_it_cannot_edit(tmp_path):
    """A rayn revision that spells the constants differently must stop the script at the edit check, not later with a size mismatch."""
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference checkout needed")
    import shutil
    fake = tmp_path / "rayn"
    shutil.copytree("/root/reference/src", fake / "src")
    p = fake / "src" / "setup.rs"
    p.write_text(p.read_text().replace("pub const SAMPLES: usize = 2;", "pub const SAMPLES : usize = 2;"))
    r = subprocess.run(["sh", os.path.join(ROOT, "tools", "pin_against_rayn.sh"), str(fake), "48", "32", "1", "2"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, MOCK_RAYN="same", PYTHON=sys.executable), cwd="/tmp")
    assert r.returncode == 4 and "could not set RESOLUTION / SAMPLES / MAX_INDIRECT_BOUNCES" in r.stderr
