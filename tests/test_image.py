"""N3: Film::save_to's per-pixel post-process (src/film.rs:205-378, src/spectrum.rs:30-40).
rayn_amd/image.py (host product code) against the oracle's arm-by-arm restatement, 8-bit exact, on an oracle-rendered film
and on adversarial values (NaN, inf, negative, > 1, values straddling quantisation steps)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from common import case
from rayn_amd import image


def _oracle_pixels(oracle, kind, film, have, transparent):
    L = oracle.lib()
    h, w = film["alpha"].shape
    L.oracle_save_to_pixels.restype = C.c_int
    fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    out = np.zeros(h * w * 4, np.uint8)
    bpp = L.oracle_save_to_pixels(C.c_uint32(kind), int(have["color"]), int(have["alpha"]), int(have["background"]), int(have["normal"]), int(transparent),
                                  C.c_uint32(w), C.c_uint32(h), fp(film["color"]), fp(film["alpha"]), fp(film["background"]), fp(film["normal"]),
                                  out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return None if bpp < 0 else out[: h * w * bpp].reshape(h, w, bpp)


def _films(oracle):
    wd, p = case("s2", 40, 24, 2, 3)
    tabs = oracle.build_tables(8, 3, p.volume_marches, p.frame, 40, 24)
    rendered, _ = oracle.render(wd, p, tabs)
    rng = np.random.default_rng(3)
    h, w = 16, 32
    adv = {"color": rng.uniform(-0.2, 1.4, (h, w, 3)).astype(np.float32), "alpha": rng.uniform(-0.1, 1.1, (h, w)).astype(np.float32),
           "background": rng.uniform(0.0, 0.5, (h, w, 3)).astype(np.float32), "normal": rng.uniform(-1.2, 1.2, (h, w, 3)).astype(np.float32)}
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 1e-30, 3.0e38, -1e-10], np.float32)
    adv["color"].reshape(-1)[: special.size] = special
    adv["alpha"].reshape(-1)[: special.size] = special
    adv["background"].reshape(-1)[: special.size] = special[::-1]
    adv["normal"].reshape(-1)[: special.size] = special
    # values whose gamma-corrected image sits next to an 8-bit step: (k / 255) ** 2.2 and its float neighbours
    k = np.arange(1, 255, dtype=np.float64)
    edge = ((k / 255.0) ** 2.2).astype(np.float32)
    edge = np.concatenate([edge, np.nextafter(edge, np.float32(0)), np.nextafter(edge, np.float32(2))])
    adv["color"].reshape(-1)[20: 20 + edge.size] = edge
    adv["alpha"].reshape(-1)[20: 20 + 254] = (k / 255.0).astype(np.float32)
    return [rendered, adv]


ALL = {"color": True, "alpha": True, "background": True, "normal": True}


def test_save_to_arms_match_the_oracle(oracle):
    for film in _films(oracle):
        # Color arms: RGBA with alpha; Color + Background composite; Color alone (no saturate)
        assert np.array_equal(image.color_image(film["color"], alpha=film["alpha"], transparent_background=True), _oracle_pixels(oracle, 0, film, ALL, True))
        assert np.array_equal(image.color_image(film["color"], background=film["background"]), _oracle_pixels(oracle, 0, film, ALL, False))
        no_bg = dict(ALL, background=False)
        assert np.array_equal(image.color_image(film["color"]), _oracle_pixels(oracle, 0, film, no_bg, False))
        assert np.array_equal(image.background_image(film["background"]), _oracle_pixels(oracle, 2, film, ALL, False))
        assert np.array_equal(image.normal_image(film["normal"]), _oracle_pixels(oracle, 3, film, ALL, False))
        assert np.array_equal(image.alpha_image(film["alpha"]), _oracle_pixels(oracle, 1, film, ALL, False))


def test_save_to_error_arms(oracle):
    film = _films(oracle)[1]
    assert _oracle_pixels(oracle, 0, film, dict(ALL, alpha=False), True) is None   # transparent without Alpha: insufficient channels
    assert _oracle_pixels(oracle, 0, film, dict(ALL, color=False), False) is None
    assert _oracle_pixels(oracle, 2, film, dict(ALL, background=False), False) is None
    with pytest.raises(ValueError):
        image.color_image(film["color"], transparent_background=True)


def test_png_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for c in (1, 3, 4):
        img = rng.integers(0, 256, (7, 11, c), dtype=np.uint8)
        path = tmp_path / f"t{c}.png"
        image.save(str(path), img)
        data = path.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        w, h, depth, ctype = struct.unpack(">IIBB", data[16:26])
        assert (w, h, depth, ctype) == (11, 7, 8, {1: 0, 3: 2, 4: 6}[c])
        pos, idat = 8, b""
        while pos < len(data):
            n, tag = struct.unpack(">I4s", data[pos:pos + 8])
            if tag == b"IDAT":
                idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        raw = zlib.decompress(idat)
        rows = np.frombuffer(raw, np.uint8).reshape(7, 1 + 11 * c)
        assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:].reshape(7, 11, c), img)


@pytest.mark.gpu
def test_film_save_to_writes_the_reference_images(tmp_path, oracle):
    """Film.save_to end to end on a GPU render: every channel arm + the Err arms."""
    import rayn_amd as R
    from rayn_amd import setup as S
    W, H = 48, 32
    cam, world = S.setup((W, H), volumes=False)
    film = R.Film([R.ChannelKind.Color, R.ChannelKind.Alpha, R.ChannelKind.Background, R.ChannelKind.WorldNormal], (W, H))
    film.render_frame_into(world, cam, R.PathTracingIntegrator(max_bounces=2, volume_marches=2), R.BlackmanHarrisFilter(1.5), (16, 16), 1, None, 1)
    kinds = [R.ChannelKind.Color, R.ChannelKind.Alpha, R.ChannelKind.Background, R.ChannelKind.WorldNormal]
    film.save_to(kinds, str(tmp_path), "t", False)
    film.save_to([R.ChannelKind.Color], str(tmp_path / "rgba"), "t", True)
    for name in ("t_color.png", "t_alpha.png", "t_background.png", "t_normal.png", "rgba/t_color.png"):
        assert (tmp_path / name).stat().st_size > 100
    arrays = {"color": film.channel(R.ChannelKind.Color), "alpha": film.channel(R.ChannelKind.Alpha),
              "background": film.channel(R.ChannelKind.Background), "normal": film.channel(R.ChannelKind.WorldNormal)}
    assert np.array_equal(image.color_image(arrays["color"], background=arrays["background"]), _oracle_pixels(oracle, 0, arrays, ALL, False))
    only_color = R.Film([R.ChannelKind.Color], (W, H))
    only_color.channels = {"color": film.channels["color"]}
    with pytest.raises(ValueError):
        only_color.save_to([R.ChannelKind.Color], str(tmp_path), "u", True)
    with pytest.raises(ValueError):
        only_color.save_to([R.ChannelKind.Alpha], str(tmp_path), "u", False)
    only_color.save_to([R.ChannelKind.Color], str(tmp_path), "u", False)
