"""Multi-device context of the C ABI (rayn_hip_create_multi): one ctx over several GPUs of the process,
tiles dealt in rotation, ONE peer copy of each device's pixels to devices[0].  On a one-GPU box the device list repeats GPU 0
(the entries then share it) - the partition, the per-entry renderers, the table broadcast, the pack / peer-copy / scatter
path and the statistics are all exercised; only the physical xGMI hop is not."""
import ctypes as C

import numpy as np
import pytest

from common import case, film_equal_bits


def test_create_multi_argument_errors():
    """No GPU needed: bad arguments are errors, and without a GPU creation fails loudly (no fallback)."""
    from rayn_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rayn_hip_create_multi(None, 1, C.byref(h)) == -1
    ids = (C.c_int * 2)(0, 0)
    assert L.rayn_hip_create_multi(ids, 0, C.byref(h)) == -1
    assert L.rayn_hip_create_multi(ids, 2, None) == -1
    assert L.rayn_hip_device_count(None) == 0
    import torch
    if not torch.cuda.is_available():
        assert L.rayn_hip_create_multi(ids, 2, C.byref(h)) != 0 and not h.value


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("name,w,h,samples,bounces,kw", [("s1", 64, 48, 2, 3, {}), ("s2", 50, 37, 1, 3, {}), ("s1", 64, 48, 1, 2, {"tile_first": 1, "tile_step": 3})])
def test_multi_ctx_matches_single_ctx(gpu_ctx, oracle, devices, name, w, h, samples, bounces, kw):
    import rayn_amd
    wd, p = case(name, w, h, samples, bounces, **kw)
    tabs = oracle.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, w, h)
    gpu_ctx.upload_world(wd)
    ref = gpu_ctx.render_host(p, tabs)
    st_ref = gpu_ctx.stats()
    m = rayn_amd.Context(devices)
    try:
        assert m.device_count() == len(devices)
        m.upload_world(wd)
        out = m.render_host(p, tabs)
        st = m.stats()
    finally:
        m.close()
    assert film_equal_bits(out, ref)
    for k in ("paths", "segments", "tiles", "shadow_jobs"):
        assert st[k] == st_ref[k], k


@pytest.mark.gpu
def test_multi_ctx_against_oracle_and_device_buffers(oracle):
    """Two entries, device-resident buffers (the bench path), volume scene, against the CPU oracle; un-owned pixels untouched."""
    import torch
    import rayn_amd
    W, H = 80, 64
    wd, p = case("s2", W, H, 2, 3)
    tabs = oracle.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, W, H)
    ref, ctr = oracle.render(wd, p, tabs)
    m = rayn_amd.Context([0, 0])
    try:
        m.upload_world(wd)
        d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
        film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
        m.render_device(p, d_tabs, film)
        torch.cuda.synchronize()
        st = m.stats()
        out = {"color": film["color"].cpu().numpy().reshape(H, W, 3), "alpha": film["alpha"].cpu().numpy().reshape(H, W),
               "background": film["background"].cpu().numpy().reshape(H, W, 3), "normal": film["normal"].cpu().numpy().reshape(H, W, 3)}
        assert st["paths"] == ctr.paths and st["segments"] == ctr.segments
        assert film_equal_bits(out, ref)
        # a tile subset on a multi ctx: only those tiles are written
        film2 = {k: torch.full_like(v, 7.0) for k, v in film.items()}
        m.set_tile_subset([1, 6, 11])
        m.render_device(p, d_tabs, film2)
        torch.cuda.synchronize()
        m.set_tile_subset(None)
        a = film2["alpha"].cpu().numpy().reshape(H, W)
        ny = (H + H % 16) // 16
        owned = np.zeros((H, W), bool)
        for k in (1, 6, 11):
            tx, ty = k // ny, k % ny
            owned[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = True
        assert np.all(a[~owned] == 7.0) and np.array_equal(a[owned].view(np.uint32), ref["alpha"][owned].view(np.uint32))
    finally:
        m.close()


@pytest.mark.gpu
def test_tables_are_broadcast_once_per_key(oracle):
    """r6: the peers' copies of the sample tables are made when the frame's (table buffers, resolution, samples, bounces, volume marches, frame) change - not
    every frame.  Three entries = two peers: two peer copies for the first frame, none for a repeat, two more after another frame number (other tables in the
    SAME device buffers would go unnoticed: the host then re-uploads the world, which forgets the broadcast), every film equal to the oracle's."""
    import torch
    import rayn_amd
    W, H = 64, 48
    wd, p = case("s1", W, H, 2, 3)
    m = rayn_amd.Context([0, 0, 0])
    try:
        m.upload_world(wd)
        film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")

        def frame(p, d_tabs, ref):
            m.render_device(p, d_tabs, film)
            torch.cuda.synchronize()
            out = {"color": film["color"].cpu().numpy().reshape(H, W, 3), "alpha": film["alpha"].cpu().numpy().reshape(H, W),
                   "background": film["background"].cpu().numpy().reshape(H, W, 3), "normal": film["normal"].cpu().numpy().reshape(H, W, 3)}
            assert film_equal_bits(out, ref)

        tabs = oracle.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, W, H)
        ref, _ = oracle.render(wd, p, tabs)
        d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
        assert m.table_broadcasts() == 0
        frame(p, d_tabs, ref)
        assert m.table_broadcasts() == 2
        frame(p, d_tabs, ref)
        frame(p, d_tabs, ref)
        assert m.table_broadcasts() == 2
        p2 = case("s1", W, H, 2, 3, frame=5)[1]
        tabs2 = oracle.build_tables(4 * p2.samples, p2.max_bounces, p2.volume_marches, p2.frame, W, H)
        ref2, _ = oracle.render(wd, p2, tabs2)
        for dst, src in zip(d_tabs, tabs2):  # the new frame's tables in the SAME device buffers: the frame number is part of the key
            dst.copy_(torch.from_numpy(src))
        frame(p2, d_tabs, ref2)
        assert m.table_broadcasts() == 4
        for dst, src in zip(d_tabs, tabs):  # rewritten in place under an unchanged key: the documented way to say so is another upload_world
            dst.copy_(torch.from_numpy(src))
        p2.frame = p.frame
        m.upload_world(wd)
        frame(p2, d_tabs, ref)
        assert m.table_broadcasts() == 6
    finally:
        m.close()
