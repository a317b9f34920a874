"""The N>1 film path on CPU: world_size-2 gloo, tiles dealt round-robin, ONE gather to rank 0
(rayn_amd/distributed.py).  The per-rank renderer here is the CPU oracle — on the GPU box the same
FilmGather runs on CUDA tensors over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import case, film_equal_bits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as O
    from rayn_amd.distributed import FilmGather
    W, H, samples, bounces = 50, 37, 1, 2  # ragged: partial tiles on both axes
    wd, p = case("s1", W, H, samples, bounces, tile_first=rank, tile_step=world)
    tabs = O.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
    local, _ = O.render(wd, p, tabs, threads=2)
    film = {k: torch.from_numpy(np.ascontiguousarray(v)).reshape(W * H, -1).squeeze(-1) for k, v in local.items()}
    g = FilmGather(W, H, (p.tile_w, p.tile_h), rank, world, "cpu")
    out = g.gather(film)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "gathered.npz"), **{k: v.numpy() for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partition_and_gather(tmp_path, oracle, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = dict(np.load(os.path.join(str(tmp_path), "gathered.npz")))
    W, H = 50, 37
    wd, p = case("s1", W, H, 1, 2)
    ref, _ = oracle.render(wd, p, oracle.build_tables(4, 2, p.volume_marches, p.frame, W, H))
    got = {"color": got["color"].reshape(H, W, 3), "alpha": got["alpha"].reshape(H, W),
           "background": got["background"].reshape(H, W, 3), "normal": got["normal"].reshape(H, W, 3)}
    assert film_equal_bits(got, ref)


def test_ownership_is_a_partition():
    from rayn_amd.distributed import owned_pixels, tile_rects
    for (w, h, world) in [(64, 48, 2), (50, 37, 8), (1920, 1080, 8)]:
        parts = [owned_pixels(w, h, 16, 16, r, world) for r in range(world)]
        allpx = np.concatenate(parts)
        # the reference's tile loop under-covers some sizes ((res + res%tile)/tile, src/film.rs:399-404)
        covered = sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tile_rects(w, h, 16, 16))
        assert len(allpx) == covered and len(np.unique(allpx)) == covered
        assert covered == w * h or (w, h) == (50, 37)
        if (w, h) == (1920, 1080):
            sizes = [len(x) for x in parts]
            assert (max(sizes) - min(sizes)) / max(sizes) < 0.04  # the half-height last tile row lands on 2 of 8 ranks


def _force_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rayn_amd.distributed import FilmGather, owned_pixels
    W, H = 50, 37
    rng = np.random.default_rng(7)  # same film on every rank: each contributes ITS pixels of it
    full = {"color": rng.random((W * H, 3), np.float32), "alpha": rng.random(W * H, np.float32),
            "background": rng.random((W * H, 3), np.float32), "normal": rng.random((W * H, 3), np.float32)}
    mine = owned_pixels(W, H, 16, 16, rank, world)
    film = {k: torch.zeros_like(torch.from_numpy(v)) for k, v in full.items()}
    for k, v in full.items():
        film[k][mine] = torch.from_numpy(v)[mine]
    g = FilmGather(W, H, (16, 16), rank, world, "cpu", force=True)
    if rank == 0:  # force: rank 0's own block travels through the collective too - it is WIPED from the film right after it has been
        # packed, so the assertions below hold only if the scatter of block 0 brings it back
        keep = {k: v.clone() for k, v in film.items()}
        orig_pack = g.pack

        def pack_then_wipe(f):
            buf = orig_pack(f)
            for k in f:
                f[k][mine] = 0
            return buf
        g.pack = pack_then_wipe
    out = g.gather(film)
    if rank == 0:
        covered = np.concatenate([owned_pixels(W, H, 16, 16, r, world) for r in range(world)])
        for k, v in full.items():
            assert np.array_equal(out[k].numpy()[covered], v[covered]), k
            assert np.array_equal(out[k].numpy()[mine], keep[k].numpy()[mine])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_forced_gather_runs_the_collective_even_at_world_1(world):
    """FilmGather(force=True) - the mode tests/test_bench_multirank.py uses to run the RCCL path on a one-GPU box."""
    mp.spawn(_force_worker, args=(world, _free_port()), nprocs=world, join=True)
