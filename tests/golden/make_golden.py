#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json FROM THE ORACLE (oracle/rayn_oracle.cpp).

The reference (fu5ha/rayn) has no tests, known-answer vectors or fixtures and cannot be built in this
environment (no Rust toolchain), so these are oracle self-goldens: they freeze the oracle's behaviour
(regression detection) and give a one-line comparison target for anyone who can run real rayn with
the same scene — they do NOT pin the oracle to rayn.  Run:  python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from common import case  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

FILMS = [("s0", 32, 32, 2, 4), ("s1", 32, 24, 1, 3), ("s2", 24, 16, 1, 2), ("s1", 20, 18, 1, 8)]


def main():
    O.build()
    out = {"films": [], "dist": {}}
    for scene, w, h, samples, bounces in FILMS:
        wd, p = case(scene, w, h, samples, bounces)
        tabs = O.build_tables(4 * samples, bounces, p.volume_marches, p.frame, w, h)
        film, ctr = O.render(wd, p, tabs, threads=2)
        hsh = hashlib.sha256()
        for k in ("color", "alpha", "background", "normal"):
            hsh.update(np.ascontiguousarray(film[k]).tobytes())
        out["films"].append({"scene": scene, "w": w, "h": h, "samples": samples, "bounces": bounces, "paths": ctr.paths,
                             "segments": ctr.segments, "packets": ctr.packets, "sha256": hsh.hexdigest(),
                             "color_mean": float(film["color"].mean())})
    wd, _ = case("s1", 16, 16, 1, 1)
    pts = [[0.0, 0.0, 0.0], [1.0, 2.0, 3.0], [-1.0125, 0.45, 4.5], [0.3, -0.7, 1.1], [2.5, 2.5, 2.5], [0.01, 0.02, -0.03],
           [100.0, -50.0, 25.0], [1.0, 1.0, 1.0]]
    d = O.sdf_dist(wd.hitables[1], np.array(pts, np.float32))
    out["dist"] = {"points": pts, "bits": [int(x) for x in d.view(np.uint32)], "values": [float(x) for x in d]}
    json.dump(out, open(os.path.join(HERE, "oracle_golden.json"), "w"), indent=1)
    print(json.dumps(out["dist"]["values"]))


if __name__ == "__main__":
    main()
