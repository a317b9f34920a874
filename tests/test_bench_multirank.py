"""bench.py's N>1 launch contract on ONE GPU: two ranks under torch.distributed.run share cuda:0 (RCCL refuses that, so the
exchange runs over gloo with the gather staged through host memory - bench.py's test aids), every other line of the rank
logic is the one the driver's 8-GPU launch executes: tile partition by tile_first / tile_step, FilmGather, barriers, max over
ranks, one JSON line from rank 0.  --check-film makes rank 0 re-render the whole frame alone and compare bit for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_two_ranks_one_gpu(ranks):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "0",
           "--workload", "c1", "--backend", "gloo", "--share-gpu", "--check-film", "--cpu-seconds", "0", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 1 and out["scaling"] == "strong"
    assert out["film_check"] is True
    assert out["value"] > 0 and out["config"]["paths_per_step"] == 256 * 256 * 16
    # the all-reduced segment count of the partitioned frame == the count of the whole frame (c1 is deterministic)
    assert out["segments_per_step"] == json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))["c1"]["frame_counts"]["segments"]
    _check_per_rank(out, ranks)
    assert sum(out["per_rank"]["tiles"]) == 256


def _check_per_rank(out, ranks):
    """The diagnosis keys of an N>1 line: every rank's HIP-event render time, wall time, gather time, segments, batches; rank 0's
    gather time; the render imbalance max / mean."""
    pr = out["per_rank"]
    for k in ("render_ms", "render_wall_ms", "gather_ms", "segments", "batches", "tiles"):
        assert len(pr[k]) == ranks, k
    assert all(v > 0 for v in pr["render_ms"]) and all(w >= 0.5 * r for w, r in zip(pr["render_wall_ms"], pr["render_ms"]))
    assert sum(pr["segments"]) == out["segments_per_step"] and all(b >= 1 for b in pr["batches"])
    assert out["gather_ms"] == pr["gather_ms"][0] and out["gather_ms"] > 0
    assert 1.0 <= out["imbalance"] < 10.0
    # r6: plot-ready keys - every rank's render time over the slowest rank's, the gather's share of a step
    assert len(pr["frac_of_slowest"]) == ranks and max(pr["frac_of_slowest"]) == 1.0 and all(0 < v <= 1.0 for v in pr["frac_of_slowest"])
    assert 0 < out["gather_share_of_step"] < 1.0 and abs(out["gather_share_of_step"] - out["gather_ms"] / out["ms_per_step"]) < 1e-3
    # the ranks' own clocks must explain the line: slowest render + its gather <= the max-over-ranks step time (+ slack for the barrier)
    assert max(pr["render_wall_ms"]) <= out["ms_per_step"] * 1.05 + 1.0


@pytest.mark.gpu
def test_bench_gather_only_two_ranks():
    """--gather-only: the frame is rendered once, then 20 gathers are timed alone (barrier before each) - the exchange separated
    from the render skew.  Two ranks on the one GPU over gloo (the xGMI hop itself needs the driver's 8-GPU node)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c1", "--backend", "gloo", "--share-gpu", "--gather-only"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["mode"] == "gather-only" and out["n_gpus"] == 2 and out["iterations"] == 20
    assert len(out["rank0_ms"]) == 20 and len(out["per_rank_mean_ms"]) == 2 and all(v > 0 for v in out["rank0_ms"])
    assert sum(out["pixels_per_rank"]) == 256 * 256 and out["bytes_per_rank"] == max(out["pixels_per_rank"]) * 40


@pytest.mark.gpu
def test_bench_rccl_code_path_world1():
    """The process-group code path of the N>1 launch on the box's REAL RCCL: `torch.distributed.run --nproc-per-node=1 bench.py
    --backend nccl --force-dist` initialises ProcessGroupNCCL with device_id, runs the barriers, the all-reduces (max time, summed
    segment count) and FilmGather's render-into-the-send-block -> dist.gather -> k_unpack_tiles (rank 0's own block included), then compares the gathered film
    with a plain single-context render bit for bit.  What a one-GPU box cannot exercise is only the xGMI hop itself."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--workload", "c1", "--backend", "nccl", "--force-dist", "--check-film", "--cpu-seconds", "0", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["film_check"] is True
    assert "RCCL gather" in out["config"]["parallelism"] and "TEST MODE" not in out["config"]["parallelism"]
    assert out["segments_per_step"] > out["config"]["paths_per_step"]  # summed over ranks by all-reduce: every path has >= 1 segment
    _check_per_rank(out, 1)


@pytest.mark.gpu
def test_bench_line_contract_n1():
    """The default (N = 1) launch on a small workload: ONE JSON line with the driver's keys, a live `roofline` (dominant march kernel,
    HIP-event timed), `roofline_hbm`, `cpu_baseline` (kind "port", bounded sample), `cold_ms` (fresh context -> first frame through
    the host-buffer entry) and the build variant "product"."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "small", "--cpu-seconds", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "cold_ms", "cold_detail"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert out["value"] > 0 and abs(out["value"] - out["config"]["paths_per_step"] / out["ms_per_step"] / 1e3) < 1e-2 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["bound_unit"] == "valu" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "k_extend1" in rf["kernel"] or "k_shadow1" in rf["kernel"] or "k_shade_setup" in rf["kernel"]
    # the flop of an evaluation comes from COUNTED iterations: the MandelBox always runs its 12 folds -> exactly 33 x 12 + 8
    assert rf["sdf"] == "mandelbox" and rf["flop_per_dist_eval"] == 404.0 and rf["sdf_iterations"] == 12 * rf["dist_evals"]
    # the whole frame (all time, march flop only) cannot beat the best of its march kernels
    assert 0 < rf["whole_frame"]["tflops"] <= max(v["tflops"] for v in rf["all_march_kernels"].values()) + 1e-6
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "tiles" in cb["sample"]
    assert out["cold_ms"] > 0 and out["config"]["build_variant"] == "product" and out["config"]["resolve_kernel"] == "k_resolve_reg<1>"
    assert abs(out["cold_detail"]["context_ms"] + out["cold_detail"]["first_frame_ms"] - out["cold_ms"]) < 0.2
    assert "per_rank" not in out  # single-process line: no process group
    assert all(0 <= v["frac"] < 1.5 for v in out["roofline_hbm"]["kernels"].values())


@pytest.mark.gpu
def test_bench_bulb_prices_the_mandelbulb_by_counted_iterations():
    """--workload bulb (Mandelbulb extension): the roofline's flop per evaluation comes from the orbit steps the instrumented kernels COUNTED -
    below the 8 x 84 + 10 = 682 of eight full steps (orbits escape at |w|^2 > 256), above one step."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    r = subprocess.run([sys.executable, "-c", "import bench, sys; bench.WORKLOADS['bulb'] = ('bulb', 240, 136, 4, 3, 'small Mandelbulb (test)'); "
                        "sys.argv = ['bench.py', '--workload', 'bulb', '--steps', '1', '--warmup', '1', '--cpu-seconds', '0', '--no-cold']; bench.main()"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rf = out["roofline"]
    assert rf["sdf"] == "mandelbulb" and 84.0 + 10.0 <= rf["flop_per_dist_eval"] < 682.0
    assert rf["dist_evals"] <= rf["sdf_iterations"] <= 8 * rf["dist_evals"]
    # r6: the scene's shadow segments are marched by k_shadow_bulb (march_bulb.h); the line carries the occupancy of its two stages and the elision accounting
    occ = rf["bulb_stage_occupancy"]["k_shadow_bulb"]
    assert 0.3 < occ["orbit"] <= 1.0 and 0.3 < occ["epilogue"] <= 1.0
    el = rf["zero_throughput_elision"]
    assert el["zero_throughput_slots"] > 0 and el["shadow_jobs_marched"] > 0 and 0 < el["share_of_segments"] < 0.5
    if "k_shadow" in rf["kernel"]:
        assert "k_shadow_bulb" in rf["kernel"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_bench_self_launch_without_a_launcher(ranks):
    """`python bench.py --gpus N` with WORLD_SIZE unset relaunches itself under torch.distributed.run (one rank per GPU) instead of exiting:
    the same single JSON line, the partitioned frame bit-identical to a single-rank render.  ranks = 8 is the driver's largest launch (here all
    eight share the one GPU over gloo: the partition, the eight packed blocks and rank 0's seven unpack launches are the real ones)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--workload", "c1", "--backend", "gloo",
           "--share-gpu", "--check-film", "--cpu-seconds", "0", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["film_check"] is True and "relaunching" in r.stderr
    _check_per_rank(out, ranks)


@pytest.mark.gpu
@pytest.mark.parametrize("entries", [2, 3])
def test_bench_single_process_multi_device(entries):
    """--single-process: ONE process, one rayn_hip_create_multi context over N entries (here all on GPU 0: --share-gpu = device list [0, 0(, 0)]) - the
    path a rayn host binds for src/film.rs:630-658.  Same JSON line; per-device render times from rayn_hip_get_entry_stats; the film is compared bit for
    bit with a plain single-device context's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(entries), "--single-process", "--share-gpu", "--steps", "2", "--warmup", "1", "--workload", "c1",
           "--check-film", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == entries and out["film_check"] is True and "rayn_hip_create_multi" in out["config"]["parallelism"]
    assert "per_rank" not in out and out["roofline"] is None and out["cpu_baseline"] is None
    pd = out["per_device"]
    assert len(pd["render_ms"]) == entries and all(v > 0 for v in pd["render_ms"]) and sum(pd["tiles"]) == 256
    assert sum(pd["segments"]) == out["segments_per_step"] == json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))["c1"]["frame_counts"]["segments"]
    assert max(pd["render_ms"]) <= out["ms_per_step"] * 1.05 + 1.0 and out["exchange_ms"] > -1.0
    assert max(pd["frac_of_slowest"]) == 1.0 and abs(out["exchange_share_of_step"] - out["exchange_ms"] / out["ms_per_step"]) < 1e-3


@pytest.mark.gpu
def test_bench_default_line_carries_the_named_workload():
    """The default workload's line (c3: the reference's fractal at the metric's size) also reports the metric's LITERALLY named workload (bulb3: the
    Mandelbulb extension, same size / volume / tables), measured on the same context after the timed region.  Both shrunk here to test size."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RAYN_HIP_ENV_TUNING="1", RAYN_HIP_BATCH_PATHS=str(1 << 22))
    code = ("import bench, sys; bench.WORKLOADS['c3'] = ('s2', 160, 96, 4, 3, 'small c3 (test)'); bench.WORKLOADS['bulb3'] = ('bulbv', 160, 96, 4, 3, 'small bulb3 (test)'); "
            "sys.argv = ['bench.py', '--steps', '1', '--warmup', '1', '--cpu-seconds', '0', '--no-cold']; bench.main()")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    nw = out["named_workload"]
    assert "error" not in nw, nw
    assert nw["workload"] == "small bulb3 (test)" and nw["value"] > 0 and nw["steps"] == 2 and nw["segments_per_step"] >= out["config"]["paths_per_step"]
    assert out["config"]["workload"] == "small c3 (test)" and out["roofline"]["sdf"] == "mandelbox"
    # r6: the named workload carries its OWN roofline - dominant kernel, fraction, counted flop per evaluation of ITS SDF - and kernel times
    nrf = nw["roofline"]
    assert nrf["sdf"] == "mandelbulb" and nrf["bound"] in ("hbm", "mfma") and 0 < nrf["frac"] < 1 and 94.0 <= nrf["flop_per_dist_eval"] < 682.0
    # (a 160x96 frame is mostly endgame: 0.28 of the orbit lane slots do work at two steps per trip, against 0.79 on the real bulb3)
    occ = nrf["bulb_stage_occupancy"]["k_shadow_bulb"]
    assert 0.1 < occ["orbit"] <= 1.0 and 0.1 < occ["epilogue"] <= 1.0 and nw["kernel_ms"]["ms_shadow"] > 0
    assert out["roofline"]["bulb_stage_occupancy"] is None  # the MandelBox frame runs k_shadow1
    # and the secondary measurement is off where it does not belong
    code2 = code.replace("'--no-cold']", "'--no-cold', '--no-named']")
    r2 = subprocess.run([sys.executable, "-c", code2], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "named_workload" not in json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
