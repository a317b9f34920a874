"""Film::save_to post-process (src/film.rs:205-378): saturate, gamma 2.2, Color+Background
composite, y-flip, 8-bit quantise.  Host-side, after the hot path.  PNG is written
with zlib only (no image crate equivalent needed)."""
import struct
import zlib

import numpy as np


def _png(path, rgb8):
    h, w, c = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ctype = {1: 0, 3: 2, 4: 6}[c]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _quant(x):
    return np.clip(x * np.float32(255.0), 0.0, 255.0).astype(np.uint8)


def color_image(color, background=None):
    """(col + bg).saturated().gamma_corrected(2.2), rows flipped (src/film.rs:247-263)."""
    c = np.asarray(color, np.float32)
    if background is not None:
        c = c + np.asarray(background, np.float32)
    c = np.clip(np.nan_to_num(c, nan=0.0), 0.0, 1.0) ** np.float32(1.0 / 2.2)
    return _quant(c)[::-1]


def normal_image(normal):
    """vec*0.5 + 0.5 (src/film.rs:324-347)."""
    return _quant(np.asarray(normal, np.float32) * np.float32(0.5) + np.float32(0.5))[::-1]


def save_color(path, color, background=None):
    _png(path, np.ascontiguousarray(color_image(color, background)))


def save_normal(path, normal):
    _png(path, np.ascontiguousarray(normal_image(normal)))
