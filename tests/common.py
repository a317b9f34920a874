"""Shared scene/config helpers for the tests (BASELINE.json configs scaled to oracle-sized cases)."""
import numpy as np

from rayn_amd import params as P
from rayn_amd import setup as S


def case(name, width, height, samples, bounces, **kw):
    """Returns (world_desc, frame_params).  name: s0 (sphere SDF), s1 (MandelBox), s2 (MandelBox + volume), s3 (MandelBox, moving camera), ship (setup::setup() as shipped = s2)."""
    cam, world = S.SCENES[name]((width, height))
    return world.to_desc(cam), P.frame_params(width, height, samples, bounces, **kw)


def film_l2(a, b):
    """max over pixels of the per-pixel L2 distance over all 10 film floats (the north_star metric,
    applied to every channel)."""
    d2 = ((a["color"].astype(np.float64) - b["color"]) ** 2).sum(-1) + ((a["background"].astype(np.float64) - b["background"]) ** 2).sum(-1) \
        + ((a["normal"].astype(np.float64) - b["normal"]) ** 2).sum(-1) + (a["alpha"].astype(np.float64) - b["alpha"]) ** 2
    return float(np.sqrt(d2).max())


def bits_equal(x, y):
    """Bit equality of float32 arrays; two NaNs compare equal whatever their sign/payload (IEEE 754 leaves the
    NaN an invalid operation produces unspecified: x86 makes 0xFFC00000, gfx950 0x7FC00000)."""
    x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
    both_nan = np.isnan(x) & np.isnan(y)
    return x.shape == y.shape and bool(np.all((x.view(np.uint32) == y.view(np.uint32)) | both_nan))


def film_equal_bits(a, b):
    return all(bits_equal(a[k], b[k]) for k in ("color", "alpha", "background", "normal"))
