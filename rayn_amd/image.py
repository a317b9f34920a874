"""Film::save_to's per-pixel post-process (src/film.rs:205-378): saturate, gamma 2.2, Color+Background composite,
Color+Alpha RGBA, normal -> RGB, y-flip, 8-bit quantise.  Host-side, after the hot path.  PNG is written with
zlib only (the reference uses the `image` crate; the encoding is not part of the path).

Arithmetic follows the reference: f32 throughout, `Srgb::saturated` = x.max(0).min(1) and `gamma_corrected(2.2)` =
x.powf(1/2.2) (src/spectrum.rs:30-40), quantisation `(v * 255.0).min(255.0).max(0.0) as u8` with Rust's f32::min/max
(a NaN operand yields the other operand) and the truncating, saturating cast.  powf is evaluated in float64 and rounded to
float32 - the same "double-evaluated, correctly rounded" result the pinned dm_powf gives (tests compare with the oracle)."""
import struct
import zlib

import numpy as np

F32 = np.float32


def _png(path, img8):
    h, w, c = img8.shape
    raw = b"".join(b"\x00" + img8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ctype = {1: 0, 3: 2, 4: 6}[c]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _rust_min(a, b):
    """f32::min: a NaN operand yields the other one."""
    with np.errstate(invalid="ignore"):
        return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b))).astype(F32)


def _rust_max(a, b):
    with np.errstate(invalid="ignore"):
        return np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b))).astype(F32)


def _quant(x):
    """(v * 255.0).min(255.0).max(0.0) as u8"""
    with np.errstate(invalid="ignore", over="ignore"):
        v = np.asarray(x, F32) * F32(255.0)
    return _rust_max(_rust_min(v, F32(255.0)), F32(0.0)).astype(np.uint8)  # truncation toward zero of a value in [0, 255]


def saturated(x):
    """Srgb::saturated, src/spectrum.rs:35-40"""
    return _rust_min(_rust_max(np.asarray(x, F32), F32(0.0)), F32(1.0))


def gamma_corrected(x, gamma=2.2):
    """Srgb::gamma_corrected, src/spectrum.rs:30-33: x.powf(1.0 / gamma) in f32 (libm conventions: negative base -> NaN,
    pow(0, y > 0) = 0, pow(1, y) = 1)."""
    x = np.asarray(x, F32)
    e = F32(1.0) / F32(gamma)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        return np.power(x.astype(np.float64), np.float64(e)).astype(F32)


def color_image(color, background=None, alpha=None, transparent_background=False):
    """The Color arm of save_to (src/film.rs:222-289).  Returns the 8-bit image (rows top-down), RGB or RGBA:
      Color + Alpha, transparent_background      -> RGBA: color.saturated().gamma_corrected(2.2) | alpha          (:231-254)
      Color + Background, not transparent        -> RGB:  (color + background).saturated().gamma_corrected(2.2)   (:255-277)
      Color only, not transparent                -> RGB:  color.gamma_corrected(2.2) - NOT saturated              (:278-ff)
    anything else is the reference's Err("... insufficient channels")."""
    c = np.asarray(color, F32)
    if transparent_background:
        if alpha is None:
            raise ValueError("Attempted to write Color channel with insufficient channels")
        rgb = _quant(gamma_corrected(saturated(c)))
        return np.concatenate([rgb, _quant(np.asarray(alpha, F32))[..., None]], axis=-1)[::-1]
    if background is not None:
        with np.errstate(invalid="ignore", over="ignore"):
            s = (c + np.asarray(background, F32)).astype(F32)
        return _quant(gamma_corrected(saturated(s)))[::-1]
    return _quant(gamma_corrected(c))[::-1]


def background_image(background):
    """saturated().gamma_corrected(2.2) (src/film.rs:290-313)"""
    return _quant(gamma_corrected(saturated(background)))[::-1]


def normal_image(normal):
    """vec * 0.5 + 0.5 (src/film.rs:314-338)"""
    with np.errstate(invalid="ignore", over="ignore"):
        return _quant(np.asarray(normal, F32) * F32(0.5) + F32(0.5))[::-1]


def alpha_image(alpha):
    """grey image of the Alpha channel (src/film.rs:339-362)"""
    return _quant(alpha)[::-1][..., None]


def save(path, img8):
    _png(path, np.ascontiguousarray(img8))
