#!/usr/bin/env python3
"""Oracle digests of WHOLE 16x16 tiles at every BASELINE configuration, under both mul_add policies where it matters.

Packet grouping is per tile (src/hitable.rs:94-134, src/film.rs:456-529), so a 16x16 tile at 1024 / 4096 spp is a
different computation from the 4x4-tile cases of FILM_CASES.  This script runs the CPU oracle on

    c1:  256x256,     16 spp,  4 bounces, sphere SDF (BASELINE configs[0]) - ALL 256 tiles = the whole frame
    c2: 1920x1080,   256 spp,  8 bounces, MandelBox                        - >= 8 spread tiles
    c3: 1920x1080,  1024 spp,  8 bounces, MandelBox + homogeneous volume   (the bench default)
    c4: 3840x2160,  1024 spp, 12 bounces, MandelBox
    c5: 7680x4320,  4096 spp, 16 bounces, MandelBox, moving camera + moving fractal (time-sampled motion blur)
    bulb / bulb3 / bulb4 / bulb5: the sizes of c2 / c3 / c4 / c5 with the power-8 Mandelbulb EXTENSION (the fractal BASELINE.json names)
    c1_fma, c3_fma: the same tiles as c1 / c3 with FUSED mul_add (librayn_oracle_fma.so = rayn built with +fma)

at FULL resolution and tile size, and writes per tile: the path / segment / packet / SDF-evaluation counts and one
SHA-256 per film channel over the tile's float32 pixels (film row order, NaNs canonicalised to 0x7FC00000).
tests/test_config_digests.py renders exactly these tiles on the GPU and compares.

Tile choice is deterministic: a 4-spp probe of ~200 spread tiles ranks them by segment count; the script takes the
most and the least expensive tile (fractal-heavy / sky), the last tile of a column (half height at 1080 rows), the
first tile, and evenly spaced quantiles of the ranking of the tiles that see more than sky.

    python tests/golden/make_config_digests.py [c1 c2 c3 c4 c5 c1_fma c3_fma] [--tiles 12] [--jobs N]      (writes config_digests.json)

About 15 core-minutes for c3 and bulb3, 5 for c4, 40 for c5, seconds for c1 (one tile runs serially on one thread, like the reference).
Every entry is stamped with `oracle_hash` = sha256 of the sources that define the oracle's arithmetic (oracle/rayn_oracle.cpp,
include/rayn_detmath.h, include/rayn_hip.h): tests/test_config_digests.py warns when the stamp no longer matches the tree and
re-derives a cheap part of every entry on the CPU, so a fixture that predates an oracle change cannot go unnoticed.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # name: (scene, W, H, samples (= spp / 4), bounces)
    "c1": ("s0", 256, 256, 4, 4),
    "c2": ("s1", 1920, 1080, 64, 8),
    "c3": ("s2", 1920, 1080, 256, 8),
    "c4": ("s1", 3840, 2160, 256, 12),
    "c5": ("s3", 7680, 4320, 1024, 16),
    # the fractal BASELINE.json names (extension scene, rayn_amd/setup.py::setup_bulb) at configs[1]'s size
    "bulb": ("bulb", 1920, 1080, 64, 8),
    # the metric's literally named workload (BASELINE.json "1920x1080 Mandelbulb @1024spp" = configs[2] with the Mandelbulb extension):
    # 1024 spp, 8 bounces, Mandelbulb + homogeneous volume (rho_s 0.25, rho_t 0.035).  The reference has no Mandelbulb (src/sdf.rs:104-141 is
    # its only fractal), so nothing in rayn corresponds to these digests: they pin the HIP path to the oracle's restated extension.
    "bulb3": ("bulbv", 1920, 1080, 256, 8),
    # BASELINE configs[3] / configs[4] WITH THE FRACTAL THEY NAME: "3840x2160, 1024 spp, 12 bounces, Mandelbulb SDF" and
    # "7680x4320, 4096 spp, 16 bounces, animated Mandelbulb with time-sampled motion blur" (moving camera + the bulb translating through
    # rayn_hitable.center_vel).  Nothing in rayn corresponds to either: its only fractal is the MandelBox (src/sdf.rs:104-141) and its TracedSDF
    # ignores time (src/sdf.rs:25,59) - c4 / c5 above are these sizes on the reference's own fractal.
    "bulb4": ("bulb", 3840, 2160, 256, 12),
    "bulb5": ("bulbm", 7680, 4320, 1024, 16),
    # the reference's OWN workload: src/main.rs:47-82 (1280x720, SAMPLES = 2 -> 8 spp, 3 bounces, frame 1, 16x16 tiles) on
    # src/setup.rs:46-170 as shipped (volumes on) - the WHOLE frame (3 600 tiles, 7.37 M paths)
    "shipped": ("ship", 1280, 720, 2, 3),
}
# fused-policy entries: (base config whose tile list is reused)
FMA_CONFIGS = {"c1_fma": "c1", "c3_fma": "c3", "shipped_fma": "shipped"}
# whole-frame entries that are too many tiles to list one by one: digests per tile COLUMN (the tile list is x-major,
# src/film.rs:399-427, so a column is a run of consecutive tiles) + one digest of the whole film
COLUMN_CONFIGS = ("shipped",)
CHANNELS = ("color", "alpha", "background", "normal")
# 8K entries: an oracle call allocates a 1.3 GB film, so few run at a time and the CPU test leaves their re-derivation to the GPU test
BIG_FILM_CONFIGS = ("c5", "bulb5")
# entries of an EXTENSION scene carry this note (the test asserts it)
EXTENSION_NOTE = ("EXTENSION: nothing in rayn corresponds to these digests - its only fractal is the MandelBox (src/sdf.rs:104-141) and its TracedSDF "
                  "ignores time (src/sdf.rs:25,59); they pin the HIP path to the oracle's restated Mandelbulb at this BASELINE size")
EXTENSION_CONFIGS = ("bulb", "bulb3", "bulb4", "bulb5")


def _code_only(text):
    """C/C++ source without comments and with whitespace collapsed: the stamp follows the ARITHMETIC, not the prose
    (a citation added to a comment must not make every fixture look stale).  A small scanner, not a regex: `//` or `/*` inside a
    string or character literal is code, not a comment."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":  # string / character literal: copy verbatim up to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
            out.append(" ")
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def oracle_hash():
    """sha256 over the comment-stripped sources that define the oracle's arithmetic."""
    h = hashlib.sha256()
    for f in ("oracle/rayn_oracle.cpp", "include/rayn_detmath.h", "include/rayn_hip.h"):
        h.update(_code_only(open(os.path.join(ROOT, f), "r").read()).encode())
    return h.hexdigest()[:16]


def base_config(name):
    return FMA_CONFIGS.get(name, name)


def is_fma(name):
    return name in FMA_CONFIGS


def world_and_params(name, samples=None):
    """The scene + frame parameters of a config (shared with the GPU test)."""
    from rayn_amd import params as P
    from rayn_amd import setup as S
    scene, W, H, smp, bounces = CONFIGS[base_config(name)]
    # s3 = config 5: moving camera (reference-supported closure) + moving fractal (TracedSDF transform_seq extension)
    cam, world = S.SCENES[scene]((W, H))
    p = P.frame_params(W, H, smp if samples is None else samples, bounces)
    return world.to_desc(cam), p


def tile_rect(p, k):
    ny = (p.height + p.height % p.tile_h) // p.tile_h
    tx, ty = k // ny, k % ny
    return tx * p.tile_w, ty * p.tile_h, min(tx * p.tile_w + p.tile_w, p.width), min(ty * p.tile_h + p.tile_h, p.height)


def tile_digests(film, p, k):
    """SHA-256 per channel over the tile's pixels (rows y0..y1, columns x0..x1 of the bottom-up film)."""
    return rect_digests(film, *tile_rect(p, k))


def rect_digests(film, x0, y0, x1, y1):
    out = {}
    for ch in CHANNELS:
        a = np.ascontiguousarray(film[ch][y0:y1, x0:x1], np.float32).copy()
        bits = a.view(np.uint32)
        bits[np.isnan(a)] = 0x7FC00000
        out[ch] = hashlib.sha256(bits.tobytes()).hexdigest()
    return out


def column_digests(film, p):
    """Whole-frame entry: one digest set per tile column (tiles tx * ny .. tx * ny + ny - 1) and one over the whole film."""
    nx = (p.width + p.width % p.tile_w) // p.tile_w
    cols = [{"column": tx, "x": [tx * p.tile_w, min(tx * p.tile_w + p.tile_w, p.width)],
             "sha256": rect_digests(film, tx * p.tile_w, 0, min(tx * p.tile_w + p.tile_w, p.width), p.height)} for tx in range(nx)]
    return cols, rect_digests(film, 0, 0, p.width, p.height)


def choose_tiles(name, n_pick, jobs, O):
    wd, p = world_and_params(name, samples=1)
    if name == "c1":  # the whole frame
        return list(range((p.width // p.tile_w) * (p.height // p.tile_h)))
    tabs = O.build_tables(4, p.max_bounces, p.volume_marches, p.frame, p.width, p.height)
    nx = (p.width + p.width % p.tile_w) // p.tile_w
    ny = (p.height + p.height % p.tile_h) // p.tile_h
    n_tiles = nx * ny
    probe = sorted(set(int(v) for v in np.linspace(0, n_tiles - 1, 200)))

    def cost(k):
        _, ctr = O.render(wd, p, tabs, threads=1, tile_subset=[k])
        return ctr.segments, ctr.paths

    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(cost, probe))
    seg = [r[0] for r in res]
    ranked = [k for _, k in sorted(zip(seg, probe))]
    busy = [k for (s_, n_), k in sorted(zip(res, probe)) if s_ > n_]  # tiles that see more than sky (some path goes beyond depth 0)
    picks = [ranked[-1], ranked[0], 0, (nx // 3) * ny + ny - 1]  # fractal-heavy, sky, first tile, last (half-height at H = 1080) tile of a column
    for q in np.linspace(0.0, 0.97, max(n_pick - len(picks), 0)):  # the rest: evenly spaced quantiles of the non-sky ranking
        picks.append(busy[int(q * (len(busy) - 1))] if busy else ranked[int(q * (len(ranked) - 1))])
    out = []
    for k in picks:
        if k not in out:
            out.append(k)
    return out


def rederive_cheap_part(name, c, O, threads=None):
    """Re-derives with the oracle AS IT IS NOW what is affordable of a fixture entry and asserts it: the whole c1 frame, one tile
    column of a column entry (45 tiles), the cheapest listed tile of every other entry (a sky tile: one segment per path).  The
    expensive tiles are re-derived by the GPU tests (the kernels share no code with the oracle but the math header)."""
    fma = is_fma(name)
    wd, p = world_and_params(name)
    tabs = O.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height, fma=fma)
    if base_config(name) == "c1":
        film, ctr = O.render(wd, p, tabs, fma=fma, threads=threads)
        assert {k: getattr(ctr, k) for k in ("paths", "segments", "packets", "dist_evals")} == c["frame_counts"]
        for t in c["tiles"]:
            assert tile_digests(film, p, t["tile"]) == t["sha256"], (name, t["tile"])
    elif base_config(name) in COLUMN_CONFIGS:
        ny = (p.height + p.height % p.tile_h) // p.tile_h
        col = c["columns"][len(c["columns"]) // 2 - 3]  # a column through the fractal
        tx = col["column"]
        film, ctr = O.render(wd, p, tabs, fma=fma, threads=threads, tile_subset=list(range(tx * ny, tx * ny + ny)))
        assert ctr.tiles == ny
        assert rect_digests(film, col["x"][0], 0, col["x"][1], p.height) == col["sha256"], (name, tx)
    elif base_config(name) not in BIG_FILM_CONFIGS:  # an 8K oracle call allocates a 1.3 GB film: left to the GPU test
        t = min(c["tiles"], key=lambda t: (t["segments"], t["paths"]))
        film, ctr = O.render(wd, p, tabs, threads=1, tile_subset=[t["tile"]], fma=fma)
        assert (ctr.paths, ctr.segments) == (t["paths"], t["segments"])
        assert tile_digests(film, p, t["tile"]) == t["sha256"], (name, t["tile"])


def rederive_expensive_part(name, c, O, jobs=1):
    """--restamp only: beyond the cheap part, re-derive with the oracle AS IT IS NOW the most expensive listed tile of every tile-sampled
    entry (fractal-heavy: several segments per path; a minute or two of one core at 1024 spp) - for c5 (a 1.3 GB film per oracle call, ~4 minutes per
    tile) the listed tile nearest the median cost - and two more tile columns of a column entry.  Returns what was verified (recorded in the entry)."""
    fma = is_fma(name)
    base = base_config(name)
    if base == "c1":
        return {"whole_frame": True}  # rederive_cheap_part already renders every tile of c1
    wd, p = world_and_params(name)
    tabs = O.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height, fma=fma)
    if base in COLUMN_CONFIGS:
        ny = (p.height + p.height % p.tile_h) // p.tile_h
        cols = [c["columns"][len(c["columns"]) // 2 + 2], c["columns"][len(c["columns"]) // 3]]
        for col in cols:
            tx = col["column"]
            film, _ = O.render(wd, p, tabs, fma=fma, threads=jobs, tile_subset=list(range(tx * ny, tx * ny + ny)))
            assert rect_digests(film, col["x"][0], 0, col["x"][1], p.height) == col["sha256"], (name, tx)
        return {"columns": [col["column"] for col in cols]}
    ranked = sorted(c["tiles"], key=lambda t: t["segments"])
    t = ranked[len(ranked) // 2] if base in BIG_FILM_CONFIGS else ranked[-1]
    film, ctr = O.render(wd, p, tabs, threads=1, tile_subset=[t["tile"]], fma=fma)
    assert (ctr.paths, ctr.segments, ctr.dist_evals) == (t["paths"], t["segments"], t["dist_evals"]), (name, t["tile"])
    assert tile_digests(film, p, t["tile"]) == t["sha256"], (name, t["tile"])
    return {"tiles": [t["tile"]], "segments": t["segments"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c1", "c2", "c3", "c4", "c5", "bulb", "bulb3", "shipped", "c1_fma", "c3_fma", "shipped_fma"])
    ap.add_argument("--tiles", type=int, default=12)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=os.path.join(HERE, "config_digests.json"))
    ap.add_argument("--restamp", action="store_true", help="after an oracle edit that did not change its arithmetic: re-derive the cheap part of every entry "
                    "(as tests/test_config_digests.py does) and, only if every digest still matches, write the current oracle_hash into the entries")
    args = ap.parse_args()
    from oracle import oracle_py as O
    O.build()
    result = json.load(open(args.out)) if os.path.exists(args.out) else {}
    if args.restamp:
        # an entry only gets the new stamp after its cheap part AND one expensive tile (two more columns of a column entry) have been
        # re-derived with the oracle as it is now; what was re-derived is recorded in the entry ("restamp_verified")
        def restamp(item):
            name, c = item
            rederive_cheap_part(name, c, O, threads=1 if base_config(name) != "c1" and base_config(name) not in COLUMN_CONFIGS else None)
            return name, rederive_expensive_part(name, c, O)
        with ThreadPoolExecutor(max(1, min(args.jobs, len(result)))) as ex:
            for name, verified in ex.map(restamp, list(result.items())):
                result[name]["oracle_hash"] = oracle_hash()
                result[name]["restamp_verified"] = verified
                print("restamped", name, verified, flush=True)
        json.dump(result, open(args.out, "w"), indent=1)
        return
    for name in args.configs:
        t0 = time.time()
        fma = is_fma(name)
        if base_config(name) in COLUMN_CONFIGS:  # the whole frame in one oracle call, all threads
            wd, p = world_and_params(name)
            tabs = O.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height, fma=fma)
            film, ctr = O.render(wd, p, tabs, threads=args.jobs, fma=fma)
            cols, whole = column_digests(film, p)
            scene, W, H, smp, bounces = CONFIGS[base_config(name)]
            result[name] = {"scene": scene, "width": W, "height": H, "samples": smp, "spp": 4 * smp, "max_bounces": bounces,
                            "volume_marches": p.volume_marches, "frame": p.frame, "tile": [p.tile_w, p.tile_h], "fma_policy": int(fma),
                            "columns": cols, "frame_sha256": whole,
                            "frame_counts": {k: getattr(ctr, k) for k in ("paths", "segments", "packets", "dist_evals", "tiles")},
                            "alpha_mean": float(film["alpha"].mean()),
                            "oracle": "oracle/rayn_oracle.cpp, " + ("FUSED mul_add (librayn_oracle_fma.so)" if fma else "unfused mul_add (librayn_oracle.so)"),
                            "oracle_hash": oracle_hash(), "seconds": round(time.time() - t0, 1)}
            json.dump(result, open(args.out, "w"), indent=1)
            print(name, "done in", round(time.time() - t0, 1), "s:", result[name]["frame_counts"], flush=True)
            continue
        # a fused entry re-renders the tile list of its base config (same tiles under both policies)
        tiles = [t["tile"] for t in result[base_config(name)]["tiles"]] if fma else choose_tiles(name, args.tiles, args.jobs, O)
        wd, p = world_and_params(name)
        tabs = O.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height, fma=fma)

        def run(k):
            film, ctr = O.render(wd, p, tabs, threads=1, tile_subset=[k], fma=fma)
            x0, y0, x1, y1 = tile_rect(p, k)
            return {"tile": k, "rect": [x0, y0, x1, y1], "paths": ctr.paths, "segments": ctr.segments, "packets": ctr.packets,
                    "dist_evals": ctr.dist_evals, "alpha_mean": float(film["alpha"][y0:y1, x0:x1].mean()), "sha256": tile_digests(film, p, k)}

        with ThreadPoolExecutor(min(args.jobs, 4 if name in BIG_FILM_CONFIGS else args.jobs)) as ex:  # 8K: 1.3 GB of film per oracle call
            recs = list(ex.map(run, tiles))
        scene, W, H, smp, bounces = CONFIGS[base_config(name)]
        result[name] = {"scene": scene, "width": W, "height": H, "samples": smp, "spp": 4 * smp, "max_bounces": bounces,
                        "volume_marches": p.volume_marches, "frame": p.frame, "tile": [p.tile_w, p.tile_h], "fma_policy": int(fma), "tiles": recs,
                        "oracle": "oracle/rayn_oracle.cpp, " + ("FUSED mul_add (librayn_oracle_fma.so)" if fma else "unfused mul_add (librayn_oracle.so)"),
                        "oracle_hash": oracle_hash(), "seconds": round(time.time() - t0, 1)}
        if base_config(name) in EXTENSION_CONFIGS:
            result[name]["note"] = EXTENSION_NOTE
        if base_config(name) == "c1":  # every tile of the frame is listed: whole-frame counts
            result[name]["frame_counts"] = {k: sum(r[k] for r in recs) for k in ("paths", "segments", "packets", "dist_evals")}
        json.dump(result, open(args.out, "w"), indent=1)
        print(name, "done in", round(time.time() - t0, 1), "s:", [(r["tile"], r["segments"]) for r in recs], flush=True)


if __name__ == "__main__":
    main()
