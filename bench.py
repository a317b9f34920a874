#!/usr/bin/env python3
"""bench.py — Mpath-samples/s of the rayn hot path on MI355X (BASELINE.json metric).

A "step" = one Film::render_frame_into of the workload (every tile, every sample, every bounce,
film resolve included) with the sample tables, scramble, filter table and scene already resident in
HBM.  The default workload is the configuration BASELINE.json's metric is quoted on ("1920x1080 ... @1024spp"
= configs[2]: 1920x1080, 1024 spp, 8 bounces, the reference's SDF fractal + homogeneous volume; the fractal is
a MandelBox — the reference has no Mandelbulb); it fits one GPU (2.12 G paths, rendered in
tile batches).  --workload c2 is configs[1] (256 spp, volumes off), bulb the added Mandelbulb DE.  N>1: the SAME
frame is partitioned by tiles (rotating round-robin) across ranks and gathered to rank 0 with ONE RCCL gather
(preallocated buffers, per-rank blocks padded to the largest share) inside the timed region -> strong scaling.

  python bench.py --gpus 1 --steps 2 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N ...                    (WORLD_SIZE unset: relaunches itself under torch.distributed.run, one rank per GPU)
  python bench.py --gpus N --single-process ...   (ONE process, one rayn_hip_create_multi context over N GPUs, peer-copy gather:
                                                   what a rayn host bound per bindings/ runs for src/film.rs:630-658)

Prints ONE JSON line on rank 0 (contract in the task statement) incl. "roofline" (dominant kernel
class, HIP-event timed live) and "cpu_baseline" (the CPU oracle on a bounded tile sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC is the only mode the host driver of the GPU boxes supports (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def usable_cpus():
    """CPUs this process may actually use: min(visible CPUs, scheduler affinity, cgroup quota).  The GPU boxes show 256 hardware threads
    but run the container under cpu.max = 16 CPUs (measured, profiles/r04_exp_workers_cold_cpu.txt: the oracle peaks at 16 threads and
    loses a third of its rate at 256) - rounds 1-3 reported `cores: 256` for what were 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def kernel_source_hash():
    """sha256 over the kernel sources, the shared math header and the build flags: PMC traffic files under profiles/ are
    stamped with it, so a stale file is detected."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "device_core.h", "march_bulb.h", "kernels.h", "device_scene.h", "rayn_hip.hip", "Makefile", "../../include/rayn_detmath.h", "../../include/rayn_detmath_fast.h",
              "../../include/rayn_logtab.h"):
        h.update(open(os.path.join(ROOT, "rayn_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


# Flop of ONE evaluation of a scene's SDF = per_iteration x (fold / orbit iterations the evaluation ran) + per_evaluation, mul = add = div = sqrt = 1
# (the reference's default build has no fma).  Iterations are COUNTED by the instrumented kernels (rayn_hip_get_sdf_iterations), not assumed.
#   MandelBox::dist (src/sdf.rs:125-188): per fold iteration 3 clamps (min + max: 6) + 3 mul_add (6) for the box fold, 5 for r^2, max + div + max (3) and 4
#     multiplies of the sphere fold, 4 mul_add (8) + a negation for scale + offset = 33; epilogue |p| (6) + abs + div = 8  ->  12 x 33 + 8 = 404
#   Mandelbulb (EXTENSION, device_core.h::mandelbulb_dist, polynomial power-8 form): per orbit step m^2, m^4 (2), dz (6), the six squares / fourth powers (6),
#     k3 (1), k2 = 1 / sqrt(k3^7) (8), k1 (11), k4 (2), w.x (15), w.y (7), w.z (20), |w|^2 (5), bailout compare (1) = 84; prologue |p|^2 (5) +
#     epilogue 0.25 ln(m) sqrt(m) / dz (5, the logarithm counted as ONE) = 10  ->  at most 8 x 84 + 10 = 682, less when the orbit escapes early
#   sdfu::Sphere: |p| - r = 7, no iterations
SDF_FLOPS = {"mandelbox": (33.0, 8.0), "mandelbulb": (84.0, 10.0), "sphere": (0.0, 7.0)}
SCENE_SDF = {"s0": "sphere", "s1": "mandelbox", "s2": "mandelbox", "s3": "mandelbox", "ship": "mandelbox", "bulb": "mandelbulb", "bulbv": "mandelbulb", "bulbm": "mandelbulb"}
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (= f32-input MFMA peak)
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (scene, width, height, samples(=spp/4), bounces, description)
    "c2": ("s1", 1920, 1080, 64, 8, "1920x1080, 256 spp, 8 bounces, MandelBox SDF (reference fractal), volumes off [BASELINE configs[1]]"),
    "c3": ("s2", 1920, 1080, 256, 8, "1920x1080, 1024 spp, 8 bounces, MandelBox SDF + homogeneous volume [BASELINE configs[2]]"),
    "c1": ("s0", 256, 256, 4, 4, "256x256, 16 spp, 4 bounces, single-sphere SDF [BASELINE configs[0]]"),
    "c4": ("s1", 3840, 2160, 256, 12, "3840x2160, 1024 spp, 12 bounces, MandelBox SDF, volumes off [BASELINE configs[3], an 8-GPU config: 8.49 G paths]"),
    "c5": ("s3", 7680, 4320, 1024, 16, "7680x4320, 4096 spp, 16 bounces, MandelBox SDF, moving camera with time-sampled motion blur "
           "[BASELINE configs[4], an 8-GPU config: 135.9 G paths; the reference's SDF itself is not time-dependent]"),
    "bulb": ("bulb", 1920, 1080, 64, 8, "1920x1080, 256 spp, 8 bounces, power-8 Mandelbulb SDF (EXTENSION: the fractal BASELINE.json names; not in the reference), volumes off"),
    # the metric's LITERALLY named workload: configs[2]'s size with the fractal BASELINE.json names.  The reference has no Mandelbulb (its only fractal is
    # the MandelBox of src/sdf.rs:104-141), so nothing in rayn corresponds to this line: it measures the HIP path on the restated extension
    "bulb3": ("bulbv", 1920, 1080, 256, 8, "1920x1080, 1024 spp, 8 bounces, power-8 Mandelbulb SDF + homogeneous volume (rho_s 0.25, rho_t 0.035) [BASELINE metric as literally named; "
              "the Mandelbulb is an EXTENSION - the reference's only fractal is a MandelBox, src/sdf.rs:104-141, so no rayn number can correspond]"),
    # r6: BASELINE configs[3] / configs[4] WITH THE FRACTAL THEY NAME (the Mandelbulb extension; tile digests: tests/golden/config_digests.json bulb4 / bulb5).  Nothing in rayn corresponds:
    # its only fractal is the MandelBox and its TracedSDF ignores time (src/sdf.rs:25,59,104-141) - "animated" = S3's moving camera + the bulb translating through center_vel
    "bulb4": ("bulb", 3840, 2160, 256, 12, "3840x2160, 1024 spp, 12 bounces, power-8 Mandelbulb SDF (EXTENSION), volumes off [BASELINE configs[3] as literally named, an 8-GPU config: 8.49 G paths]"),
    "bulb5": ("bulbm", 7680, 4320, 1024, 16, "7680x4320, 4096 spp, 16 bounces, power-8 Mandelbulb SDF (EXTENSION) translating during the shutter, moving camera with time-sampled motion blur "
              "[BASELINE configs[4] as literally named, an 8-GPU config: 135.9 G paths]"),
    # the reference's OWN workload, the only thing rayn itself times (src/main.rs:47-82 on src/setup.rs:46-170 as shipped): here the first
    # frame of a fresh context (cold_ms) is the number that corresponds to a rayn run, and the CPU leg renders the WHOLE frame in the GPU's own 16x16 tiles
    "shipped": ("ship", 1280, 720, 2, 3, "1280x720, 8 spp (SAMPLES = 2), 3 bounces, MandelBox SDF + homogeneous volume, frame 1: the reference's shipped default "
                "(src/main.rs:47-82, src/setup.rs:16-60)"),
    "mid": ("s1", 960, 540, 16, 8, "960x540, 64 spp, 8 bounces, MandelBox (profiling-sized)"),
    "small": ("s1", 480, 270, 4, 3, "480x270, 16 spp, 3 bounces, MandelBox (quick check)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 2; 20 for the sub-100-ms workloads shipped / c1 / small)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames before them (default 1; 3 for shipped / c1 / small)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-named", action="store_true", help="default workload only: skip the secondary measurement of the metric's literally named workload (bulb3) reported as `named_workload`")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold first frame through the host-buffer entry (cold_ms)")
    ap.add_argument("--fma-policy", type=int, default=0, choices=[0, 1], help="0: unfused mul_add = rayn's default build (default); 1: fused = rayn built with +fma")
    # test aids for the N>1 path on a box with fewer GPUs than ranks (never used by the driver's launch)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo stages the gather through host memory)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks render on cuda:0 (RCCL refuses two ranks on one device: use with --backend gloo)")
    ap.add_argument("--check-film", action="store_true", help="rank 0 re-renders the whole frame alone after the timed region and compares it bit for bit with the gathered film")
    ap.add_argument("--gather-only", action="store_true", help="diagnosis of the N>1 exchange: render the frame once, then time 20 film gathers alone "
                    "(barrier before each) and print per-rank / per-iteration gather times instead of the bench line - separates the xGMI gather from render skew")
    ap.add_argument("--single-process", action="store_true", help="N GPUs inside ONE process: one rayn_hip_create_multi context over devices 0..N-1 (peer-copy film gather over "
                    "xGMI, no process group) - the path a rayn host bound per bindings/ uses for src/film.rs:630-658; same JSON line with per-device render times")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group, barrier, all-reduce, FilmGather incl. rank 0's own block) even at "
                    "world size 1: executes the RCCL path of the multi-GPU launch on a one-GPU box")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        # `python bench.py --gpus N` without a launcher: become the driver's own launch line (one rank per GPU under torch.distributed.run).
        # exec, not spawn: same stdout (ONE JSON line from rank 0), same exit code, nothing of this process left behind.
        # --standalone: the launcher's own c10d rendezvous on a port IT picks (no bind-close-reuse race between this process and the launcher)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: relaunching as `%s`\n" % (args.gpus, " ".join(cmd[1:10]) + " ..."))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    quick = args.workload in ("shipped", "c1", "small")
    if args.steps is None:
        args.steps = 20 if quick else 2
    if args.warmup is None:
        args.warmup = 3 if quick else 1

    import numpy as np
    import torch
    import torch.distributed as dist

    import rayn_amd
    from rayn_amd import setup as S
    from rayn_amd.distributed import FilmGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_devices = 1  # GPUs driven by THIS process (> 1 only with --single-process)
    if args.single_process:
        if world != 1:
            sys.exit("bench.py: --single-process is one process over N GPUs; do not launch it under torch.distributed.run")
        n_devices = args.gpus
    elif world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} disagrees with the launcher's WORLD_SIZE={world}; measuring {world} rank(s)\n")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    if args.single_process and not args.share_gpu and torch.cuda.device_count() < n_devices:
        sys.exit(f"bench.py: --single-process --gpus {n_devices} needs {n_devices} visible GPUs ({torch.cuda.device_count()} here; --share-gpu puts all entries on GPU 0)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    scene, W, H, samples, bounces, desc = WORKLOADS[args.workload]
    spp = 4 * samples
    cam, wld = S.SCENES[scene]((W, H))
    wd = wld.to_desc(cam)
    p = rayn_amd.frame_params(W, H, samples, bounces, tile_first=rank, tile_step=world)
    tabs = rayn_amd.build_tables(spp, bounces, p.volume_marches, p.frame, W, H)
    # cold_ms: what a host that renders ONE frame per process pays - the reference's own usage (src/main.rs:47-96, frame_range 1..2):
    # context creation -> world upload -> first frame through the HOST-buffer entry (tables up, film down) on a context with no
    # device memory yet.  Outside the timed region; the same context then serves the warm-up and the timed steps.
    t_cold = time.perf_counter()
    devices = ([0] * n_devices if args.share_gpu else list(range(n_devices))) if args.single_process else None
    ctx = rayn_amd.Context(devices if devices is not None else local_rank)
    ctx.upload_world(wd)
    ctx.set_fma_policy(args.fma_policy)
    cold_ms = None
    cold_detail = None
    measure_cold = not use_dist and not args.no_cold and n_devices == 1
    if not measure_cold:
        # no cold frame is measured (N > 1, --single-process, --no-cold): the context's FIRST frame is then the first warm-up frame, and the library's first-frame
        # policy (small arenas for hosts that render one frame per process, rayn_hip_set_cold_bytes) would move the one-off growth of the arenas into the SECOND
        # frame - the first timed step at --warmup 1.  Full-size arenas at once: every one-off cost lands in the warm-up.
        ctx.set_cold_bytes(0)
    if measure_cold:
        t_created = time.perf_counter()
        ctx.render_host(p, tabs)
        cold_ms = (time.perf_counter() - t_cold) * 1e3
        cold_detail = {"context_ms": round((t_created - t_cold) * 1e3, 1), "first_frame_ms": round(cold_ms - (t_created - t_cold) * 1e3, 1),
                       "note": "context_ms = rayn_hip_create + upload_world in THIS process, whose HIP runtime torch has already started (hipInit, 56-71 ms, is "
                               "not in it; the first queue, 85-104 ms, is); first_frame_ms = rayn_hip_render_frame on host buffers (tables up, arenas allocated, "
                               "code objects loaded, film down).  A torch-less host (tools/cold_frame.py) pays 200-244 ms from process start for the shipped workload"}
    d_tabs = [torch.from_numpy(t).to(device) for t in tabs]  # resident in HBM before the timed region
    film = rayn_amd.film.alloc_device_film(W, H, device)
    # ranks other than 0 resolve their share straight into the gather's packed send block (no full-resolution film, no pack step)
    gather = FilmGather(W, H, (p.tile_w, p.tile_h), rank, world, device, stage_host=(args.backend == "gloo"), force=args.force_dist, ctx=ctx, params=p) if use_dist else None

    # per-rank diagnosis of an N>1 run (all-gathered after the timed region): HIP-event time of the render (rayn_stats.ms_total, events
    # on the caller's stream around the whole share), host wall time of the render call, and the gather (pack + collective + scatter,
    # fenced with a device synchronise - the render call is blocking already, and the next frame follows the gather on the same stream,
    # so the fence costs one host round trip)
    acc = {"render_ms": 0.0, "render_wall_ms": 0.0, "gather_ms": 0.0}
    dev_acc = [{"render_ms": 0.0, "segments": 0, "batches": 0, "tiles": 0} for _ in range(n_devices)]  # --single-process: per entry of the context

    def step(timed=False):
        t_a = time.perf_counter()
        if gather is not None:
            gather.render(d_tabs, film)
        else:
            ctx.render_device(p, d_tabs, film)
        t_b = time.perf_counter()
        res = film
        if gather is not None:
            res = gather.gather(film)
            torch.cuda.synchronize()
        t_c = time.perf_counter()
        if timed and n_devices > 1:
            for e, d in enumerate(dev_acc):
                es = ctx.entry_stats(e)
                d["render_ms"] += es["ms_total"]
                d["segments"], d["batches"], d["tiles"] = es["segments"], es["batches"], es["tiles"]
        if timed:
            acc["render_ms"] += ctx.stats()["ms_total"]
            acc["render_wall_ms"] += (t_b - t_a) * 1e3
            acc["gather_ms"] += (t_c - t_b) * 1e3
        return res

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    xdev = device if args.backend == "nccl" else "cpu"

    def all_reduce(x, op):
        t = torch.tensor([x], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def all_gather_vec(vec):
        """[world][len(vec)] float64: every rank's vector on every rank (one all_gather)."""
        mine = torch.tensor(vec, dtype=torch.float64, device=xdev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return [[float(v) for v in t.tolist()] for t in parts]

    if args.gather_only:
        if not use_dist:
            sys.exit("bench.py: --gather-only needs the process-group path (N > 1 ranks, or --force-dist)")
        step()
        iters = 20
        mine = []
        for _ in range(iters):
            fence()
            t_a = time.perf_counter()
            gather.gather(film)
            torch.cuda.synchronize()
            mine.append((time.perf_counter() - t_a) * 1e3)
        allr = all_gather_vec(mine)
        if rank == 0:
            print(json.dumps({"mode": "gather-only", "metric": "film gather (ONE dist.gather of the ranks' packed planar blocks + one k_unpack_tiles launch per block on rank 0; "
                              "no pack step: ranks resolve into their send block), ms", "n_gpus": world, "iterations": iters,
                              "backend": args.backend, "bytes_per_rank": gather.block * 40, "pixels_per_rank": gather.counts,
                              "rank0_ms": [round(v, 3) for v in allr[0]], "per_rank_mean_ms": [round(sum(r) / len(r), 3) for r in allr],
                              "per_rank_min_ms": [round(min(r), 3) for r in allr], "workload": desc}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        ctx.close()
        return

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    stats = ctx.stats()
    segments_per_step = stats["segments"]
    per_rank = None
    if use_dist:
        dt = all_reduce(dt, dist.ReduceOp.MAX)
        segments_per_step = int(all_reduce(float(stats["segments"]), dist.ReduceOp.SUM))  # every rank's own share (exact below 2^53)
        rows = all_gather_vec([acc["render_ms"] / args.steps, acc["render_wall_ms"] / args.steps, acc["gather_ms"] / args.steps,
                               float(stats["segments"]), float(stats["batches"]), float(stats["tiles"])])
        per_rank = {"render_ms": [round(r[0], 3) for r in rows], "render_wall_ms": [round(r[1], 3) for r in rows],
                    "gather_ms": [round(r[2], 3) for r in rows], "segments": [int(r[3]) for r in rows], "batches": [int(r[4]) for r in rows],
                    "tiles": [int(r[5]) for r in rows]}
    film_check = None
    if (use_dist or n_devices > 1) and args.check_film and rank == 0:  # outside the timed region: the whole frame on ONE device of this rank alone
        p_full = rayn_amd.frame_params(W, H, samples, bounces)
        film_full = rayn_amd.film.alloc_device_film(W, H, device)
        if n_devices > 1:
            solo = rayn_amd.Context(local_rank)
            solo.upload_world(wd)
            solo.set_fma_policy(args.fma_policy)
            solo.render_device(p_full, d_tabs, film_full)
            solo.close()
        else:
            ctx.render_device(p_full, d_tabs, film_full)
        torch.cuda.synchronize()
        film_check = all(torch.equal(result[k].view(torch.int32), film_full[k].view(torch.int32)) for k in ("color", "alpha", "background", "normal"))
        del film_full
    paths_per_step = W * H * spp
    value = paths_per_step * args.steps / dt / 1e6

    def measure_rooflines(scene_tag, workload_tag):
        """Rooflines of the workload whose world is uploaded in ctx, outside any timed region: (1) one frame with per-kernel-class HIP events on the production kernels
        (ONE worker: with two, kernels of both streams overlap and their durations stretch), (2) one frame with the instrumented kernel variants (SDF evaluations,
        fold / orbit iterations, elision and stage accounting).  Returns (roofline, roofline_hbm, kernel_ms)."""
        # (1) per-kernel-class HIP-event timing with the production kernels, (2) SDF-evaluation counts with
        # the instrumented variants.  Both outside the timed region.
        # per-kernel event times are only meaningful without the two-worker overlap: profile with ONE worker
        ctx.set_workers(1)
        ctx.set_profiling(True, False)
        ctx.render_device(p, d_tabs, film)
        torch.cuda.synchronize()
        st = ctx.stats()
        ctx.set_profiling(False, True)
        ctx.render_device(p, d_tabs, film)
        torch.cuda.synchronize()
        ev = ctx.eval_counts()
        it = ctx.sdf_iterations()
        el = ctx.elision_counts()
        slots = ctx.stage_slots()
        st_count = ctx.stats()
        ctx.set_profiling(False, False)
        ctx.set_workers(2)
        # flop of the evaluations = per-iteration flop x the iterations the instrumented kernels COUNTED + the per-evaluation part (SDF_FLOPS):
        # exactly 404 per evaluation for the shipped MandelBox; the Mandelbulb's orbits escape early, so its average is below the 682 of 8 full steps
        sdf_kind = SCENE_SDF[scene_tag]
        f_it, f_ev = SDF_FLOPS[sdf_kind]
        flops = {k: f_it * it[k] + f_ev * ev[k] for k in ev}
        classes = {"extend": (st["ms_extend"], ev["extend"], st["launches_extend"], flops["extend"]), "shadow": (st["ms_shadow"], ev["shadow"], st["launches_shade"], flops["shadow"]),
                   "shade_setup": (st["ms_shade"], ev["shade_setup"], st["launches_shade"], flops["shade_setup"])}
        dom = max(classes, key=lambda k: classes[k][0])
        ms, evals, launches, flop = classes[dom]
        achieved = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        flop_per_eval = flop / evals if evals else 0.0
        all_flop = sum(v[3] for v in classes.values())
        frame_tflops = all_flop / (st["ms_total"] * 1e-3) / 1e12 if st["ms_total"] > 0 else 0.0
        # "bound" keeps the bench contract's two-word vocabulary (hbm | mfma = "the compute roof"); "bound_unit" names the unit that roof belongs to here:
        # the FP32 VECTOR pipe (VALU) - nothing on this path is a dense contraction and no MFMA instruction is issued
        kname = {"extend": "k_extend1", "shadow": "k_shadow_bulb" if sdf_kind == "mandelbulb" else "k_shadow1", "shade_setup": "k_shade_setup"}[dom]
        roofline = {"kernel": f"{kname} [FP32 VALU-bound; no MFMA instruction is issued anywhere on this path]", "bound": "mfma", "bound_unit": "valu",
                    "bound_detail": "compute-bound on the FP32 VALU (divergent scalar math, no MFMA issued); "
                    "peak = MI355X dense FP32 peak, 157.3 TFLOP/s for the vector pipe and for f32-input MFMA alike",
                    "achieved": round(achieved, 3), "peak": FP32_VECTOR_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4), "traffic": None,
                    "launches": launches, "avg_launch_ms": round(ms / max(launches, 1), 4),
                    "flop_per_launch": flop / max(launches, 1), "dist_evals": evals,
                    "sdf": sdf_kind, "flop_per_dist_eval": round(flop_per_eval, 2), "sdf_iterations": it[dom],
                    "flop_model": f"{f_it:g} flop per fold / orbit iteration x counted iterations + {f_ev:g} per evaluation (bench.py SDF_FLOPS; iterations and evaluations counted "
                                  "by the instrumented kernel variants outside the timed region)",
                    "whole_frame": {"tflops": round(frame_tflops, 3), "frac": round(frame_tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
                                    "note": "SDF-evaluation flop of ALL three march classes over the whole frame's HIP-event time (queue kernels, shading arithmetic and the film resolve "
                                            "count as time but not as flop)"},
                    "all_march_kernels": {k: {"ms": round(v[0], 3), "dist_evals": v[1], "flop_per_dist_eval": round(v[3] / v[1], 2) if v[1] else 0.0,
                                              "tflops": round(v[3] / max(v[0], 1e-9) / 1e9, 3)} for k, v in classes.items()},
                    "note": "march kernels are FP32-VALU bound (SURVEY.md F6): achieved = SDF-evaluation flop / kernel time; "
                            "peak = MI355X FP32 vector peak (= dense f32-input MFMA peak)",
                    # r6, single-Mandelbulb scenes (march_bulb.h): occupancy of the two stages of an evaluation = lane slots used / lane slots offered (instrumented kernels)
                    "bulb_stage_occupancy": None if not slots["shadow_orbit"] else {
                        "k_shadow_bulb": {"orbit": round(it["shadow"] / max(slots["shadow_orbit"], 1), 4), "epilogue": round(ev["shadow"] / max(slots["shadow_epilogue"], 1), 4)}},
                    # r6: shaded segments whose throughput is exactly (0, 0, 0) park no shadow segment (every NEE term is multiplied by that zero, src/integrator.rs:91-92,128-129;
                    # exact - k_shade_setup): counted by the instrumented kernels; shadow_jobs_marched + elided_shadow_jobs = what r5 marched
                    "zero_throughput_elision": {**el, "segments": st_count["segments"], "shadow_jobs_marched": st_count["shadow_jobs"],
                                                "share_of_segments": round(el["zero_throughput_slots"] / max(st_count["segments"], 1), 4),
                                                "share_of_shadow_jobs": round(el["elided_shadow_jobs"] / max(el["elided_shadow_jobs"] + st_count["shadow_jobs"], 1), 4)}}
        # HBM-bound queue kernels: algorithmic bytes / HIP-event time per kernel class
        npool = st["paths"]
        qk = {
            "k_raygen": (st["ms_raygen"], 85.0 * npool),                      # 5 records of 16 B + term_info + queue entry per path
            "bin(k_group_hist+k_scan_tile+k_tile_prefix+k_bin_scatter)": (st["ms_bin"], st["queue_bytes_bin"]),
            "compact(k_scan_tile+k_tile_prefix+k_compact_scatter)": (st["ms_compact"], st["queue_bytes_compact"]),
            "k_resolve": (st["ms_resolve"], 37.0 * npool + 40.0 * W * H / world),  # col0 + aov + termination record per path, film out
        }
        qk[("k_shadow_bulb" if sdf_kind == "mandelbulb" else "k_shadow1") + " (queue side: 4 B ref + 24 B segment in, 1 B visibility out per shadow job; the kernel itself is VALU-bound)"] = (st["ms_shadow"], 29.0 * st["shadow_jobs"])
        roofline_hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "shadow_jobs": st["shadow_jobs"], "kernels": {}}
        for name, (ms_k, nbytes) in qk.items():
            ach = nbytes / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
            roofline_hbm["kernels"][name] = {"ms": round(ms_k, 3), "algorithmic_bytes": nbytes, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
        # PMC traffic is measured by separate rocprofv3 --pmc passes (tools/gpu_round.sh) and committed under profiles/ with the
        # hash of the kernel sources it was taken on; a file that does not match the sources of THIS build is not quoted.
        pmc_name = next((n for n in (f"r06_pmc_hbm_{workload_tag}.json", f"r05_pmc_hbm_{workload_tag}.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        pmc_path = os.path.join(ROOT, "profiles", pmc_name or "")
        if pmc_name and world == 1 and args.fma_policy == 0:
            pj = json.load(open(pmc_path))
            if pj.get("source_hash") == kernel_source_hash():
                pmc = pj["kernels"]
                if kname in pmc and "hbm_bytes_per_launch" in pmc[kname]:
                    roofline["traffic"] = pmc[kname]["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, kernel sources {pj['source_hash']})"
                if kname in pmc and "valu_inst_per_cycle_simd" in pmc[kname]:
                    # same file, same kernel sources: SQ counters of the dominant kernel.  A wave64 binary32 VALU instruction issues at
                    # one per 2 cycles per SIMD (what the 157.3 TFLOP/s peak is made of: 64 lanes x 2 flop / 2 cycles x 1024 SIMDs x 2.4 GHz)
                    ipc = pmc[kname]["valu_inst_per_cycle_simd"]
                    roofline["lanes_enabled"] = round(pmc[kname]["lanes_enabled"], 4)  # SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU): the exec-masked fold block
                    roofline["valu_issue"] = {"inst_per_cycle_simd": round(ipc, 4), "peak": 0.5, "frac": round(ipc / 0.5, 4),
                                              "lanes_enabled": round(pmc[kname]["lanes_enabled"], 4),
                                              "source": f"profiles/{pmc_name} (rocprofv3 --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE passes)"}
                roofline_hbm["pmc_traffic"] = {k: {"hbm_bytes": v["hbm_bytes"], "GBps": round(v.get("hbm_GBps", 0.0), 1)} for k, v in pmc.items()}
                # k_shade_finish has no closed-form algorithmic byte count (its records exist per light-receiving hit): it is rated on
                # the MEASURED traffic of the same kernel sources (PMC pass) over the live HIP-event time of this run
                fin = next((v for k, v in pmc.items() if "k_shade_finish" in k), None)
                if fin and st["ms_finish"] > 0:
                    ach = fin["hbm_bytes"] / (st["ms_finish"] * 1e-3) / 1e9
                    roofline_hbm["kernels"]["k_shade_finish (PMC-measured HBM bytes)"] = {"ms": round(st["ms_finish"], 3), "hbm_bytes": fin["hbm_bytes"], "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
            else:
                roofline["traffic_note"] = f"profiles/{pmc_name} was measured on other kernel sources ({pj.get('source_hash')} != {kernel_source_hash()}): not quoted"
        kernel_ms = {k: round(st[k], 3) for k in ("ms_raygen", "ms_extend", "ms_bin", "ms_shade", "ms_shadow", "ms_finish", "ms_compact", "ms_resolve", "ms_total")}

        return roofline, roofline_hbm, kernel_ms

    roofline = None
    roofline_hbm = None
    kernel_ms = None
    if not args.no_roofline and n_devices == 1:
        roofline, roofline_hbm, kernel_ms = measure_rooflines(scene, args.workload)

    cpu_baseline = None
    if rank == 0 and world == 1 and n_devices == 1 and args.cpu_seconds > 0:
        from oracle import oracle_py as O
        threads = usable_cpus()  # what the container may use (cgroup quota), not the host's thread count
        # The oracle runs a tile serially on one thread (like the reference).  A 16x16 tile of this workload is about a minute
        # of one core at 1024 spp, so the CPU sample uses smaller tiles (a legal Film::render_frame_into tile_size) that hold
        # <= 4096 paths each: same scene, resolution, spp, bounces and per-path work, bounded wall time and a balanced thread
        # pool.  (Packet composition is per tile, so the CPU side's packets differ from the GPU's 16x16 tiles - stated in the
        # emitted record; parity at 16x16 is what tests/test_config_digests.py checks.)
        ct = 16
        while ct > 1 and ct * ct * spp > 4096:
            ct //= 2
        whole_frame = W * H * spp <= (1 << 23)  # the shipped workload (7.4 M paths): the CPU leg renders the WHOLE frame in the GPU's own 16x16 tiles
        p0 = rayn_amd.frame_params(W, H, samples, bounces, tile_size=(ct, ct))
        n_tiles = rayn_amd._lib.lib().rayn_tile_count(W, H, p0.tile_w, p0.tile_h)
        # calibrate on one spread tile per thread, then run whole rounds of tiles per thread for ~cpu_seconds (>= 3 rounds, so that
        # the expensive fractal tiles and the cheap sky tiles balance out over the thread pool)
        def run(k):
            sub = np.unique(np.linspace(0, n_tiles - 1, num=min(k, n_tiles)).astype(np.uint32))
            t = time.perf_counter()
            _, ctr = O.render(wd, p0, tabs, threads=threads, tile_subset=sub)
            return time.perf_counter() - t, ctr.paths, len(sub)
        if whole_frame:
            t_w = time.perf_counter()
            _, ctr_w = O.render(wd, p, tabs, threads=threads)
            t_w = time.perf_counter() - t_w
            cpu_baseline = {"value": round(ctr_w.paths / t_w / 1e6, 4), "unit": "Mpath-samples/s", "cores": threads, "host_threads_visible": os.cpu_count(), "kind": "port",
                            "tile": [p.tile_w, p.tile_h], "gpu_tile": [p.tile_w, p.tile_h], "frame_ms": round(t_w * 1e3, 1),
                            "sample": f"the WHOLE frame ({ctr_w.tiles} tiles of {p.tile_w}x{p.tile_h} pixels = the GPU's own tiles and packet grouping, {ctr_w.paths} paths, "
                                      f"{ctr_w.segments} segments), C++ oracle (restatement of rayn's CPU path; rayn itself cannot be built here), "
                                      f"{threads} threads, one tile per task like the reference's rayon pool, {t_w:.1f} s"}
        else:
            t_cal, paths_cal, k_cal = run(threads)
            rounds = int(min(128, max(3, round(args.cpu_seconds / max(t_cal, 1e-3)))))  # r5: cap 32 -> 128 rounds (a c3 sample stopped at 4.7 s of the 15 s it is given)
            k = int(min(n_tiles, threads * rounds))
            t_cpu, paths_cpu, k_used = run(k)
            cpu_baseline = {"value": round(paths_cpu / t_cpu / 1e6, 4), "unit": "Mpath-samples/s", "cores": threads, "host_threads_visible": os.cpu_count(), "kind": "port",
                            "tile": [ct, ct], "gpu_tile": [p.tile_w, p.tile_h],
                            "sample": f"{k_used} of {n_tiles} {ct}x{ct}-pixel tiles (evenly spread, {paths_cpu} paths) of the same workload, C++ oracle "
                                      f"(restatement of rayn's CPU path; rayn itself cannot be built here), {t_cpu:.1f} s; the GPU renders "
                                      f"{p.tile_w}x{p.tile_h} tiles (same per-path work, different packet grouping)"}

    # BASELINE.json words its metric "1920x1080 Mandelbulb @1024spp".  The reference has no Mandelbulb, so `value` is measured on the reference's own
    # fractal at that size (c3); the literally named workload - the Mandelbulb EXTENSION with the same volume, same size, same tables - is measured right after,
    # on the same context, OUTSIDE the timed region, and reported beside it (one warm-up + two timed frames, ~30 s).  `--workload bulb3` gives it a full line.
    named = None
    if rank == 0 and world == 1 and n_devices == 1 and args.workload == "c3" and not args.no_named and not args.no_roofline:
        try:
            scene_b, Wb, Hb, samples_b, bounces_b, desc_b = WORKLOADS["bulb3"]
            assert (Wb, Hb, samples_b, bounces_b) == (W, H, samples, bounces)  # same frame parameters and sample tables
            cam_b, wld_b = S.SCENES[scene_b]((Wb, Hb))
            ctx.upload_world(wld_b.to_desc(cam_b))
            ctx.render_device(p, d_tabs, film)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            for _ in range(2):
                ctx.render_device(p, d_tabs, film)
            torch.cuda.synchronize()
            dt_b = (time.perf_counter() - t_b) / 2
            st_b = ctx.stats()
            rf_b, rf_hbm_b, kms_b = measure_rooflines(scene_b, "bulb3")
            named = {"workload": desc_b, "value": round(paths_per_step / dt_b / 1e6, 3), "unit": "Mpath-samples/s", "ms_per_step": round(dt_b * 1e3, 3), "steps": 2, "warmup": 1,
                     "segments_per_step": st_b["segments"],
                     # its OWN roofline: dominant kernel, HIP-event time, counted flop per evaluation; lanes_enabled / traffic from this workload's PMC file
                     "roofline": {k: rf_b[k] for k in ("kernel", "bound", "bound_unit", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "dist_evals", "sdf",
                                                       "flop_per_dist_eval", "whole_frame", "bulb_stage_occupancy", "zero_throughput_elision", "lanes_enabled", "valu_issue") if k in rf_b},
                     "kernel_ms": kms_b,
                     "note": "secondary measurement on the same context after the timed region of `value` (one warm-up + two timed frames, then one profiling and one counting frame); "
                             "parity of this workload: tests/test_config_digests.py [bulb3]; cpu_baseline and queue-kernel rooflines: python bench.py --workload bulb3 (profiles/r06_bench_bulb3_*.json)"}
            ctx.upload_world(wd)
        except Exception as e:  # never lose the main line to the secondary measurement
            named = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        out = {
            "metric": "Mpath-samples/sec", "value": round(value, 3), "unit": "Mpath-samples/s", "n_gpus": world * n_devices, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "paths_per_step": paths_per_step, "tile": [p.tile_w, p.tile_h], "fma_policy": "unfused (reference default build)" if args.fma_policy == 0 else "fused (reference built with +fma)",
                       "parallelism": f"tiles round-robin over {world * n_devices} GPU(s)" +
                                      (", one RCCL gather of the ranks' packed planar blocks to rank 0 per frame (ranks resolve straight into their send block; rank 0 unpacks "
                                       "with one kernel launch per block); every frame is fenced after its gather (per-rank gather_ms)" if use_dist else "") +
                                      (f", ONE process: a rayn_hip_create_multi context over devices {devices} (one host thread + renderer per device, one hipMemcpyPeerAsync of its "
                                       "packed block per peer over xGMI, one unpack launch per block on devices[0]); no process group" if n_devices > 1 else "")},
            "roofline": roofline, "roofline_hbm": roofline_hbm, "cpu_baseline": cpu_baseline, "kernel_ms": kernel_ms,
            "segments_per_step": segments_per_step,
            "cold_ms": None if cold_ms is None else round(cold_ms, 1),  # fresh context -> first frame done (host buffers), see above
        }
        out["config"]["build_variant"] = rayn_amd._lib.build_variant() or "product"
        out["config"]["resolve_kernel"] = ("k_resolve_reg<%d>" % max(1, 1 << max(0, (spp - 1).bit_length() - 6)) if spp <= 512 else
                                           "k_resolve_blk<128, 8>" if spp <= 1024 else "k_resolve_blk<256, 8>" if spp <= 2048 else
                                           "k_resolve_blk<512, 8>" if spp <= 4096 else "k_resolve_huge")
        if named is not None:
            out["named_workload"] = named
        if cold_detail is not None:
            out["cold_detail"] = cold_detail
        if per_rank is not None:
            # what the first real multi-GPU run needs to be diagnosable: every rank's render time (HIP events around its share), its
            # gather time, its share of the work; gather_ms = rank 0's (the receiver: pack + collective + scatter of N - 1 blocks)
            # r6: two keys a driver can plot without post-processing: every rank's render time as a fraction of the slowest rank's (what the launch waits for),
            # and the share of a step that rank 0 spends in the film gather (collective + unpack, fenced)
            slowest = max(per_rank["render_ms"]) or 1.0
            per_rank["frac_of_slowest"] = [round(v / slowest, 4) for v in per_rank["render_ms"]]
            out["per_rank"] = per_rank
            out["gather_ms"] = per_rank["gather_ms"][0]
            out["gather_share_of_step"] = round(per_rank["gather_ms"][0] / max(dt / args.steps * 1e3, 1e-9), 5)
            mean_r = sum(per_rank["render_ms"]) / len(per_rank["render_ms"])
            out["imbalance"] = round(max(per_rank["render_ms"]) / mean_r, 4) if mean_r > 0 else None
        if n_devices > 1:
            # per DEVICE of the one context: HIP-event time of its own share on its own stream, its segments / batches / tiles; exchange_ms = the
            # frame on devices[0]'s clock minus the slowest device's render = table broadcast + peer copies + unpack (what the gather costs)
            rms = [d["render_ms"] / args.steps for d in dev_acc]
            out["per_device"] = {"render_ms": [round(v, 3) for v in rms], "segments": [d["segments"] for d in dev_acc], "batches": [d["batches"] for d in dev_acc],
                                 "tiles": [d["tiles"] for d in dev_acc]}
            out["per_device"]["frac_of_slowest"] = [round(v / (max(rms) or 1.0), 4) for v in rms]
            out["exchange_ms"] = round(acc["render_ms"] / args.steps - max(rms), 3)
            out["exchange_share_of_step"] = round(out["exchange_ms"] / max(dt / args.steps * 1e3, 1e-9), 5)
            mean_r = sum(rms) / len(rms)
            out["imbalance"] = round(max(rms) / mean_r, 4) if mean_r > 0 else None
        if film_check is not None:
            out["film_check"] = film_check  # gathered film == single-rank film, bit for bit
        if use_dist and args.backend != "nccl":
            out["config"]["parallelism"] += f" [TEST MODE: backend {args.backend}, gather staged through host memory" + (", all ranks on one GPU" if args.share_gpu else "") + "]"
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
