#!/usr/bin/env python3
"""Dump / compare the observable state of one render in a documented, toolchain-neutral format (SURVEY.md
section 8c "Net", row N2): the only route to ever pin this project's oracle against REAL rayn.

  python tools/rayn_dump.py dump OUT_DIR [--scene s2 --w 64 --h 64 --samples 2 --bounces 3 --tile 1 --backend oracle|gpu]
                                         [--fma 1] [--variant NAME] [--tables-from DIR]
  python tools/rayn_dump.py compare DIR_A DIR_B
  python tools/rayn_dump.py mockrayn OUT_DIR --as same|tables|fma|<variant> [--scene ship --w .. --h .. --samples .. --bounces .. --tile ..]
        a STAND-IN for the dump a patched rayn writes (bindings/rayn_dump.patch: src/dump.rs), produced by the oracle so that
        tools/pin_against_rayn.sh can be run end to end WITHOUT a Rust toolchain (MOCK_RAYN=<as>): same file names, the manifest keys
        src/dump.rs writes (backend "rayn").  `same` = rayn is the oracle's default reading; `tables` = rayn's sample tables / scramble differ
        from ours (tables of another frame seed) but its arithmetic is ours; `fma` / <variant> = rayn's arithmetic is that OTHER reading.

  --scene ship = the reference's setup::setup() as shipped; --fma 1 = the fused-mul_add oracle (rayn built with +fma);
  --variant NAME = one assumption of the oracle read the other way (oracle/SENSITIVITY.md, oracle_py.VARIANTS);
  --tables-from DIR = take samples_1d / samples_2d / scramble / fis from another dump (rayn's own): a film that still differs then
  is not the tables' fault (assumptions A6 / A7 are out of the picture).

A dump is a directory of raw little-endian arrays + manifest.json:
  samples_1d.f32   Samples::samples_1d  (spp * sets_1d)            src/sampler.rs:11-15
  samples_2d.f32   Samples::samples_2d  (2 * spp * sets_2d)
  scramble.f32     SmallRng::seed_from_u64(x + y*w).gen::<f32>() per pixel, index x + y*w    src/film.rs:460-461
  fis.f32          FilterImportanceSampler::inverse_cdf (512)      src/filter.rs:187-220
  color.f32 / alpha.f32 / background.f32 / normal.f32   Film channels after tile_finished, index x + y*w (y = 0 bottom)
  trace.u32        per-depth packet lanes of tile `tile` in HitStore::process_hits order, 6 u32 per lane:
                   depth, object id, tile x, tile y, sample, valid         src/hitable.rs:94-134
INTEGRATION.md shows the ~20 lines of Rust that write the same files from rayn.  `compare` reports, per array, the
number of differing bit patterns, the max abs difference and (for the film) the max per-pixel L2."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32 = ["samples_1d", "samples_2d", "scramble", "fis", "color", "alpha", "background", "normal"]


def dump(args):
    from common import case
    from oracle import oracle_py as O
    wd, p = case(args.scene, args.w, args.h, args.samples, args.bounces)
    fma, variant = bool(args.fma), args.variant
    tabs = O.build_tables(4 * args.samples, args.bounces, p.volume_marches, p.frame, args.w, args.h, fma=fma, variant=variant)
    if args.tables_from:
        theirs = tuple(np.fromfile(os.path.join(args.tables_from, k + ".f32"), np.float32) for k in F32[:4])
        assert all(a.shape == b.shape for a, b in zip(theirs, tabs)), "table sizes differ: other spp / bounce count / resolution?"
        tabs = theirs
    if args.backend == "gpu":
        import rayn_amd
        ctx = rayn_amd.Context(0)
        ctx.upload_world(wd)
        ctx.set_fma_policy(int(fma))
        ctx.set_trace_tile(args.tile)
        film = ctx.render_host(p, tabs)
        trace = ctx.trace()
    else:
        film, _ = O.render(wd, p, tabs, fma=fma, variant=variant)
        trace = O.trace_tile(wd, p, tabs, args.tile, fma=fma, variant=variant)
    os.makedirs(args.out, exist_ok=True)
    arrays = dict(zip(F32[:4], tabs))
    arrays.update({k: film[k] for k in ("color", "alpha", "background", "normal")})
    for k, a in arrays.items():
        np.ascontiguousarray(a, np.float32).tofile(os.path.join(args.out, k + ".f32"))
    tr = np.stack([trace[k] for k in ("depth", "obj", "px", "py", "sample", "valid")], axis=1).astype(np.uint32)
    tr.tofile(os.path.join(args.out, "trace.u32"))
    json.dump({"scene": args.scene, "width": args.w, "height": args.h, "SAMPLES": args.samples, "spp": 4 * args.samples,
               "max_bounces": args.bounces, "volume_marches": p.volume_marches, "frame": p.frame, "time_range": [p.time_start, p.time_end],
               "tile": [p.tile_w, p.tile_h], "trace_tile": args.tile, "backend": args.backend,
               "fma": int(fma), "variant": variant, "tables_from": args.tables_from}, open(os.path.join(args.out, "manifest.json"), "w"), indent=1)
    print(f"wrote {args.out}: {len(arrays) + 1} arrays")


def mockrayn(args):
    """The oracle playing rayn (see the module docstring): what tools/pin_against_rayn.sh must conclude is known in advance for every --as."""
    from common import case
    from oracle import oracle_py as O
    wd, p = case(args.scene, args.w, args.h, args.samples, args.bounces)
    as_ = args.as_
    fma = as_ == "fma"
    variant = as_ if as_ in O.VARIANTS else None
    assert as_ in ("same", "tables", "fma") or variant, f"--as {as_}: not one of same, tables, fma, {O.VARIANTS}"
    # rayn's tables are computed by rayn's OWN host code (quasi-rd, SmallRng, its filter): in the mock they come from the default
    # oracle build, except under `tables`, where they are another frame's (= "A6 / A7 read wrong, arithmetic right")
    tabs = O.build_tables(4 * args.samples, args.bounces, p.volume_marches, p.frame + (1 if as_ == "tables" else 0), args.w, args.h)
    film, _ = O.render(wd, p, tabs, fma=fma, variant=variant)
    trace = O.trace_tile(wd, p, tabs, args.tile, fma=fma, variant=variant)
    os.makedirs(args.out, exist_ok=True)
    arrays = dict(zip(F32[:4], tabs))
    arrays.update({k: film[k] for k in ("color", "alpha", "background", "normal")})
    for k, a in arrays.items():
        np.ascontiguousarray(a, np.float32).tofile(os.path.join(args.out, k + ".f32"))
    np.stack([trace[k] for k in ("depth", "obj", "px", "py", "sample", "valid")], axis=1).astype(np.uint32).tofile(os.path.join(args.out, "trace.u32"))
    # exactly the keys src/dump.rs::write_manifest writes (bindings/make_patch.py; tests/test_bindings.py keeps the two in step)
    json.dump({"scene": "rayn setup::setup() [MOCK: the oracle as '%s']" % as_, "width": args.w, "height": args.h, "SAMPLES": args.samples, "spp": 4 * args.samples,
               "max_bounces": args.bounces, "volume_marches": p.volume_marches, "frame": p.frame, "time_range": [p.time_start, p.time_end],
               "tile": [p.tile_w, p.tile_h], "trace_tile": args.tile, "backend": "rayn"}, open(os.path.join(args.out, "manifest.json"), "w"), indent=1)
    print(f"wrote {args.out}: mock rayn dump as '{as_}'")


def compare(args):
    ma, mb = (json.load(open(os.path.join(d, "manifest.json"))) for d in (args.a, args.b))
    keys = ("width", "height", "spp", "max_bounces", "volume_marches", "frame")
    if any(ma[k] != mb[k] for k in keys):
        print("manifests differ:", {k: (ma[k], mb[k]) for k in keys if ma[k] != mb[k]})
        return 2
    worst = 0
    for name in F32 + ["trace"]:
        ext, dt = (".u32", np.uint32) if name == "trace" else (".f32", np.float32)
        a, b = (np.fromfile(os.path.join(d, name + ext), dt) for d in (args.a, args.b))
        if a.shape != b.shape:
            print(f"{name:12s} SHAPE {a.shape} vs {b.shape}")
            worst = 1
            continue
        if dt is np.uint32:
            nd = int((a != b).sum())
            print(f"{name:12s} {a.size:9d} values, {nd} differ" + (f", first at lane {int(np.argmax(a != b)) // 6}" if nd else ""))
        else:
            both_nan = np.isnan(a) & np.isnan(b)
            nd = int(((a.view(np.uint32) != b.view(np.uint32)) & ~both_nan).sum())
            mad = float(np.nanmax(np.abs(a.astype(np.float64) - b))) if a.size else 0.0
            extra = ""
            if name in ("color", "background", "normal"):
                extra = f", max per-pixel L2 {float(np.sqrt(((a.astype(np.float64) - b).reshape(-1, 3) ** 2).sum(1)).max()):.3e}"
            print(f"{name:12s} {a.size:9d} values, {nd} bit patterns differ, max |a-b| {mad:.3e}{extra}")
        worst = max(worst, 1 if nd else 0)
    print("IDENTICAL" if worst == 0 else "DIFFERENT")
    return worst


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("dump")
    d.add_argument("out")
    d.add_argument("--scene", default="s2")
    d.add_argument("--w", type=int, default=64)
    d.add_argument("--h", type=int, default=64)
    d.add_argument("--samples", type=int, default=2)
    d.add_argument("--bounces", type=int, default=3)
    d.add_argument("--tile", type=int, default=1)
    d.add_argument("--backend", default="oracle", choices=["oracle", "gpu"])
    d.add_argument("--fma", type=int, default=0, choices=[0, 1])
    d.add_argument("--variant", default=None)
    d.add_argument("--tables-from", default=None)
    m = sub.add_parser("mockrayn")
    m.add_argument("out")
    m.add_argument("--as", dest="as_", required=True)
    for sp, default_scene in ((m, "ship"),):
        sp.add_argument("--scene", default=default_scene)
        sp.add_argument("--w", type=int, default=1280)
        sp.add_argument("--h", type=int, default=720)
        sp.add_argument("--samples", type=int, default=2)
        sp.add_argument("--bounces", type=int, default=3)
        sp.add_argument("--tile", type=int, default=1)
    c = sub.add_parser("compare")
    c.add_argument("a")
    c.add_argument("b")
    args = ap.parse_args()
    sys.exit({"dump": dump, "mockrayn": mockrayn, "compare": compare}[args.cmd](args))


if __name__ == "__main__":
    main()
