"""The packed planar film of a share (include/rayn_hip.h: rayn_share_pixels, rayn_hip_render_frame_packed_device,
rayn_hip_unpack_share_device) - what a rank of the multi-process launch hands to the film gather (rayn_amd/distributed.py) and what
rank 0 scatters into its film; the same layout and kernels rayn_hip_create_multi uses between the devices of one process.
No reference counterpart: tiles are independent (src/film.rs:439-627), tile_finished copies a finished tile into the one film
(src/film.rs:660-691)."""
import ctypes as C

import numpy as np
import pytest

from common import case, film_equal_bits


def test_share_pixels_is_the_size_of_the_owned_tiles():
    """No GPU needed: rayn_share_pixels == the pixel count of the share's tiles (the host mirror's owned_pixels), the shares of a
    frame partition exactly what the reference's tile loop covers (incl. its (res + res % tile) / tile under-coverage quirk,
    src/film.rs:399-404), and bad arguments give 0."""
    from rayn_amd import _lib
    from rayn_amd.distributed import owned_pixels, tile_rects
    from rayn_amd.film import share_pixels
    from rayn_amd.params import frame_params
    for (w, h, tw, th, world) in [(64, 48, 16, 16, 2), (50, 37, 16, 16, 3), (1920, 1080, 16, 16, 8), (256, 256, 16, 16, 1), (33, 20, 8, 4, 5)]:
        counts = []
        for r in range(world):
            p = frame_params(w, h, 1, 1, tile_size=(tw, th), tile_first=r, tile_step=world)
            counts.append(share_pixels(p))
            assert counts[-1] == len(owned_pixels(w, h, tw, th, r, world))
        assert sum(counts) == sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tile_rects(w, h, tw, th))
    L = _lib.lib()
    bad = frame_params(64, 48, 1, 1, tile_first=2, tile_step=2)
    assert L.rayn_share_pixels(C.byref(bad)) == 0 and L.rayn_share_pixels(None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,samples,bounces,world", [("s1", 64, 48, 2, 3, 2), ("s2", 50, 37, 1, 3, 3), ("s0", 96, 80, 1, 2, 8)])
def test_packed_shares_reassemble_the_frame(gpu_ctx, oracle, name, w, h, samples, bounces, world):
    """Every share rendered STRAIGHT into its packed planar film, then scattered into one film with rayn_hip_unpack_share_device:
    bit-identical to the whole frame rendered at once (and to the oracle); the packed planes themselves hold the owned pixels in
    the documented order (tile after tile, x-major inside a tile)."""
    import torch
    import rayn_amd
    from rayn_amd.distributed import owned_pixels
    from rayn_amd.film import share_pixels
    wd, p = case(name, w, h, samples, bounces)
    tabs = oracle.build_tables(4 * samples, bounces, p.volume_marches, p.frame, w, h)
    ref, _ = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
    whole = rayn_amd.film.alloc_device_film(w, h, "cuda:0")
    gpu_ctx.render_device(p, d_tabs, whole)
    film = rayn_amd.film.alloc_device_film(w, h, "cuda:0")
    for ch in film:
        film[ch].fill_(-7.0)  # pixels no share owns (under-covered resolutions) must stay untouched
    seg = 0
    for r in range(world):
        _, pr = case(name, w, h, samples, bounces, tile_first=r, tile_step=world)
        n = share_pixels(pr)
        packed = torch.full((10 * n + 16,), 123.0, dtype=torch.float32, device="cuda:0")
        gpu_ctx.render_packed(pr, d_tabs, packed)
        seg += gpu_ctx.stats()["segments"]
        assert bool((packed[10 * n:] == 123.0).all())  # nothing written past the share's planes
        idx = torch.from_numpy(owned_pixels(w, h, pr.tile_w, pr.tile_h, r, world)).cuda()
        assert torch.equal(packed[:3 * n].view(n, 3).view(torch.int32), whole["color"][idx].view(torch.int32))
        assert torch.equal(packed[3 * n:4 * n].view(torch.int32), whole["alpha"][idx].view(torch.int32))
        assert torch.equal(packed[4 * n:7 * n].view(n, 3).view(torch.int32), whole["background"][idx].view(torch.int32))
        assert torch.equal(packed[7 * n:10 * n].view(n, 3).view(torch.int32), whole["normal"][idx].view(torch.int32))
        gpu_ctx.unpack_share(pr, packed, film)
        gpu_ctx.unpack_share(pr, packed, film)  # second use of a share: the cached tile list, idempotent
    torch.cuda.synchronize()
    covered = torch.from_numpy(np.concatenate([owned_pixels(w, h, p.tile_w, p.tile_h, r, world) for r in range(world)])).cuda()
    mask = torch.zeros(w * h, dtype=torch.bool, device="cuda:0")
    mask[covered] = True
    for ch in film:
        assert torch.equal(film[ch][mask].view(torch.int32), whole[ch][mask].view(torch.int32)), ch
        assert bool((film[ch][~mask] == -7.0).all()), ch
    gpu_ctx.render_device(p, d_tabs, whole)
    assert seg == gpu_ctx.stats()["segments"]
    if bool(mask.all()):
        host = {"color": film["color"].cpu().numpy().reshape(h, w, 3), "alpha": film["alpha"].cpu().numpy().reshape(h, w),
                "background": film["background"].cpu().numpy().reshape(h, w, 3), "normal": film["normal"].cpu().numpy().reshape(h, w, 3)}
        assert film_equal_bits(host, ref)


@pytest.mark.gpu
def test_packed_entry_argument_errors(gpu_ctx):
    import torch
    import rayn_amd
    wd, p = case("s1", 64, 48, 1, 1)
    gpu_ctx.upload_world(wd)
    L = gpu_ctx._L
    buf = torch.zeros(64 * 48 * 10, device="cuda:0")
    assert L.rayn_hip_render_frame_packed_device(gpu_ctx.h, C.byref(p), None, None, None, None, None, None) == -1
    gpu_ctx.set_tile_subset([0, 1])
    try:
        d = C.c_void_p(buf.data_ptr())
        assert L.rayn_hip_render_frame_packed_device(gpu_ctx.h, C.byref(p), d, d, d, d, d, None) == -1
        assert b"tile subset" in L.rayn_hip_last_error(gpu_ctx.h)
    finally:
        gpu_ctx.set_tile_subset(None)
    m = rayn_amd.Context([0, 0])
    try:
        m.upload_world(wd)
        assert L.rayn_hip_render_frame_packed_device(m.h, C.byref(p), d, d, d, d, d, None) == -1
        st = m.entry_stats(0)
        assert st["paths"] == 0
        with pytest.raises(rayn_amd.film.RaynHipError):
            m.entry_stats(2)
    finally:
        m.close()
