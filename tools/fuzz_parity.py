"""Hunt for rare GPU/oracle mismatches: N seeded random scenes (the generator of tests/test_gpu_parity.py's
test_randomised_scene_parity, larger films, up to 1024 spp).  Every third scene goes through a two-entry multi-device context
(rayn_hip_create_multi on GPU 0 twice); r5: every third OTHER scene is rendered share by share (2-5 ranks' shares) straight into the packed planar
films of the multi-process gather and reassembled with rayn_hip_unpack_share_device.  r6: a third argument `bulb` makes every scene a Mandelbulb
scene (k_shadow_bulb, march_bulb.h) with random march budgets.  usage: fuzz_parity.py [n=60] [first_seed=100] [bulb]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import rayn_amd as R
from rayn_amd import setup as S, params as P
from oracle import oracle_py as O
from common import film_equal_bits, film_l2
n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
only_bulb = len(sys.argv) > 3 and sys.argv[3] == "bulb"
ctx1 = R.Context(0)
ctx2 = R.Context([0, 0])
bad = 0
t_start = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(32, 97)), int(rng.integers(24, 65))
    volumes = bool(rng.integers(0, 2))
    kind = rng.choice(["mandelbox", "mandelbox", "mandelbox", "mandelbulb", "sphere"])
    if only_bulb:
        kind = "mandelbulb"
    cam_h, world = S.setup((w, h), volumes=volumes, sdf=str(kind))
    if kind == "mandelbox":
        box = world.hitables[1].sdf
        box.iterations = int(rng.integers(4, 16))
        box.scale = float(np.float32(rng.uniform(-2.8, -1.6) if rng.integers(0, 2) else rng.uniform(1.7, 3.0)))
        box.box_fold.side_length = float(np.float32(rng.uniform(0.6, 1.5)))
        box.sphere_fold.min_radius = float(np.float32(rng.uniform(0.002, 0.7)))
        box.sphere_fold.fixed_radius = float(np.float32(rng.uniform(0.75, 2.5)))
    if kind == "mandelbox" and rng.integers(0, 4) == 0:
        world.hitables[1].sdf.scale_vel = float(np.float32(rng.uniform(-4, 4)))  # extension: morphing fractal
    if volumes:
        world.volume_params = R.VolumeParams(float(np.float32(rng.uniform(0.02, 0.8))), float(np.float32(rng.uniform(0.005, 0.3))))
    for L in world.lights:
        L.pos = L.pos + rng.uniform(-0.4, 0.4, 3).astype(np.float32)
    cam = world.cameras.get(cam_h)
    cam.origin = (cam.origin * np.float32(rng.uniform(0.4, 1.6)) + rng.uniform(-0.6, 0.6, 3).astype(np.float32)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        world.hitables[1].transform_seq = R.Linear(rng.uniform(-0.3, 0.3, 3).astype(np.float32), rng.uniform(-4, 4, 3).astype(np.float32))
    if rng.integers(0, 4) == 0:
        cam.origin = R.Linear(cam.origin, rng.uniform(-3, 3, 3).astype(np.float32))
    if only_bulb:
        world.hitables[1].sdf.iterations = int(rng.integers(1, 13))
    wd = world.to_desc(cam_h)
    samples, bounces = int(rng.choice([1, 2, 3, 4, 4, 8, 16, 33, 64, 150, 256])), int(rng.integers(0, 9))
    if samples >= 8:  # many samples per pixel (the resolve sorts 4*samples keys): keep the film small
        w, h = max(8, w // (samples // 4 + 1)), max(6, h // 4)
    t0 = float(np.float32(rng.uniform(0.0, 3.0)))
    p = P.frame_params(w, h, samples, bounces, frame=int(rng.integers(1, 200)), time_range=(t0, float(np.float32(t0 + rng.uniform(0.005, 0.3)))),
                       tile_size=(int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16, 32]))), volume_marches=int(rng.choice([2, 2, 3, 4])),
                       **({"max_marches": int(rng.choice([256, 256, 40, 3])), "max_vis_marches": int(rng.choice([100, 100, 12, 1]))} if only_bulb else {}))
    tabs = O.build_tables(4 * samples, bounces, p.volume_marches, p.frame, w, h)
    ref, ctr = O.render(wd, p, tabs)
    ctx = ctx2 if seed % 3 == 0 else ctx1
    ctx.upload_world(wd)
    if seed % 3 == 1:  # the film gather of a multi-process launch: every rank's share -> its packed film -> one unpack each
        import copy
        from rayn_amd.film import share_pixels
        world_n = int(rng.integers(2, 6))
        d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
        film = R.film.alloc_device_film(w, h, "cuda:0")
        st = {"paths": 0, "segments": 0}
        for r in range(world_n):
            pr = copy.copy(p)
            pr.tile_first, pr.tile_step = r, world_n
            packed = torch.empty(10 * max(share_pixels(pr), 1), dtype=torch.float32, device="cuda:0")
            ctx.render_packed(pr, d_tabs, packed)
            sr = ctx.stats()
            st["paths"] += sr["paths"]; st["segments"] += sr["segments"]
            ctx.unpack_share(pr, packed, film)
        torch.cuda.synchronize()
        out = {"color": film["color"].cpu().numpy().reshape(h, w, 3), "alpha": film["alpha"].cpu().numpy().reshape(h, w),
               "background": film["background"].cpu().numpy().reshape(h, w, 3), "normal": film["normal"].cpu().numpy().reshape(h, w, 3)}
    else:
        out = ctx.render_host(p, tabs)
        st = ctx.stats()
    ok = st["paths"] == ctr.paths and st["segments"] == ctr.segments and film_equal_bits(out, ref)
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}: {kind} {w}x{h} spp {4*samples} B {bounces} vol {volumes} L2 {film_l2(out, ref)} segs {st['segments']} vs {ctr.segments}", flush=True)
print(f"{n} scenes, {bad} mismatches, {time.time() - t_start:.1f} s")
