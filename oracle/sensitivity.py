#!/usr/bin/env python3
"""How much does each UNPINNED assumption matter?  (VERDICT r3 "next" 2c; CPU only.)

The oracle's header lists the third-party semantics it assumes (A1-A9).  For every assumption that can change a pixel there is a
variant library that reads it the other way (oracle/Makefile `variants`, A1 = librayn_oracle_fma.so).  This script renders, with the
default oracle and with every variant,
    c1       the whole BASELINE configs[0] frame (256x256, 16 spp, 4 bounces, sphere SDF)
    shipped  the whole frame of the reference's own workload (1280x720, 8 spp, 3 bounces, MandelBox + volume; src/main.rs:47-82)
    c3       two whole 16x16 tiles of configs[2] (1920x1080, 1024 spp, 8 bounces, volume): the most expensive digest tile and a mid one
and reports per variant the per-pixel L2 distance (Color + Background, the image north_star's 1e-4 bound is about; and all ten film
floats) against the default: max, 99.9th percentile, share of pixels beyond 1e-4, share of pixels that differ at all.  The tables of a
variant are built by the variant (the filter table uses Lerp).  Output: a markdown table (oracle/SENSITIVITY.md is this script's
output plus commentary).    usage: python oracle/sensitivity.py [--jobs N] [--skip-c3] > table.md"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_config_digests as G  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

READINGS = [  # (label, assumption, kwargs of oracle_py.render / build_tables)
    ("fused mul_add (rayn built with +fma)", "A1", {"fma": True}),
    ("max/min operands swapped", "A2", {"variant": "minmax_swapped"}),
    ("max/min = IEEE maxNum/minNum (lane-wise f32::max)", "A2", {"variant": "minmax_ieee"}),
    ("host libm expf/sinf/cosf/tanf/atan2f/powf", "A3", {"variant": "libm"}),
    ("normalized() = three divisions", "A4", {"variant": "normalize_div"}),
    ("dot = (x*x' + y*y') + z*z', no mul_add nesting", "A4", {"variant": "dot_plain"}),
    ("central-difference normals (6 evals)", "A5", {"variant": "normals_central"}),
    ("tetrahedral normals, other summation order", "A5", {"variant": "normals_order"}),
    ("lerp = a + (b - a) t", "A5", {"variant": "lerp_alt"}),
]


def l2(a, b, keys):
    d2 = 0.0
    for k in keys:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        d = np.nan_to_num(x - y, nan=0.0, posinf=0.0, neginf=0.0)
        d = d * d
        d2 = d2 + (d.sum(-1) if d.ndim == 3 else d)
    return np.sqrt(d2)


def stats(ref, got, mask=None):
    img = l2(ref, got, ("color", "background"))
    full = l2(ref, got, ("color", "background", "normal", "alpha"))
    if mask is not None:
        img, full = img[mask], full[mask]
    differ = np.zeros(img.shape, bool)
    for k in ("color", "alpha", "background", "normal"):
        ne = ref[k].view(np.uint32) != got[k].view(np.uint32)
        ne &= ~(np.isnan(ref[k]) & np.isnan(got[k]))
        ne = ne.any(-1) if ne.ndim == 3 else ne
        differ |= ne[mask] if mask is not None else ne
    return {"max": float(img.max()), "p999": float(np.quantile(img, 0.999)), "over": float((img > 1e-4).mean()), "differ": float(differ.mean()),
            "max_all": float(full.max()), "mean": float(img.mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--skip-c3", action="store_true")
    ap.add_argument("--json", default=os.path.join(HERE, "sensitivity.json"))
    args = ap.parse_args()
    O.build()
    O.build_variants()
    digests = json.load(open(os.path.join(ROOT, "tests", "golden", "config_digests.json")))
    c3_tiles = sorted(digests["c3"]["tiles"], key=lambda t: t["segments"])
    c3_pick = [c3_tiles[-1]["tile"], c3_tiles[len(c3_tiles) // 2]["tile"]]
    cases = [("c1", "c1", None), ("shipped", "shipped", None)] + ([] if args.skip_c3 else [("c3 (2 tiles)", "c3", c3_pick)])
    results = {}
    for label, name, subset in cases:
        wd, p = G.world_and_params(name)

        def run(**kw):
            tabs = O.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height, **kw)
            film, ctr = O.render(wd, p, tabs, threads=args.jobs, tile_subset=subset, **kw)
            return film, ctr
        t0 = time.time()
        ref, ctr0 = run()
        mask = None
        if subset is not None:
            mask = np.zeros((p.height, p.width), bool)
            for k in subset:
                x0, y0, x1, y1 = G.tile_rect(p, k)
                mask[y0:y1, x0:x1] = True
        results[label] = {"paths": ctr0.paths, "segments": ctr0.segments, "readings": {}}
        for rlabel, assumption, kw in READINGS:
            got, ctr = run(**kw)
            st = stats(ref, got, mask)
            st["segments_delta"] = int(ctr.segments) - int(ctr0.segments)
            results[label]["readings"][rlabel] = dict(st, assumption=assumption)
            print(f"# {label:14s} {rlabel:52s} max {st['max']:.3e} p99.9 {st['p999']:.3e} >1e-4 {100 * st['over']:.3f} % differ {100 * st['differ']:.2f} % segs {st['segments_delta']:+d}",
                  file=sys.stderr, flush=True)
        print(f"# {label}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    json.dump(results, open(args.json, "w"), indent=1)
    # markdown
    for label, res in results.items():
        print(f"\n**{label}** ({res['paths']} paths, {res['segments']} segments; per-pixel L2 over Color + Background vs the default oracle)\n")
        print("| assumption | alternative reading | max L2 | 99.9 % L2 | mean L2 | pixels > 1e-4 | pixels that differ at all | segments |")
        print("|---|---|---|---|---|---|---|---|")
        for rlabel, st in res["readings"].items():
            print(f"| {st['assumption']} | {rlabel} | {st['max']:.2e} | {st['p999']:.2e} | {st['mean']:.2e} | {100 * st['over']:.3f} % | {100 * st['differ']:.2f} % | {st['segments_delta']:+d} |")


if __name__ == "__main__":
    main()
