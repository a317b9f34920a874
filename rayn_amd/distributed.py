"""Multi-GPU film partition: one process per GPU, whole 16x16 tiles dealt round-robin to ranks
(tile k belongs to rank (k + k // world) % world — tiles are fully independent, src/film.rs:439-627, and their
cost is very uneven, so ranks are interleaved and the interleave rotates every `world` tiles), and ONE gather of each rank's owned pixels to
rank 0 at frame end (`dist.gather` = ncclGather on RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).  No other
communication touches the data path.  (Inside ONE process the C ABI does the same with peer copies: rayn_hip_create_multi.)"""
import numpy as np


def tile_rects(width, height, tile_w, tile_h):
    """The reference's tile list, x-major, incl. the (res + res%tile)/tile quirk (src/film.rs:399-427)."""
    nx, ny = (width + width % tile_w) // tile_w, (height + height % tile_h) // tile_h
    return [(tx * tile_w, ty * tile_h, min(tx * tile_w + tile_w, width), min(ty * tile_h + tile_h, height))
            for tx in range(nx) for ty in range(ny)]


def owned_pixels(width, height, tile_w, tile_h, rank, world):
    """Film pixel indices (x + y*width) of the tiles rank owns, in tile order."""
    idx = []
    for k, (x0, y0, x1, y1) in enumerate(tile_rects(width, height, tile_w, tile_h)):
        if (k + k // world) % world != rank or x1 <= x0 or y1 <= y0:
            continue
        xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), indexing="ij")
        idx.append((xs + ys * width).reshape(-1))
    return np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)


class FilmGather:
    """Index plan + PREALLOCATED buffers for gathering a tile-partitioned film onto rank 0 (nothing is allocated inside a
    frame).  The exchange is ONE collective: `dist.gather` to rank 0 (ncclGather semantics on RCCL: every rank contributes a
    buffer of the same size, so the per-rank pixel blocks are padded to the largest share - shares differ by < 4 %: the
    half-height last tile row).  Rank 0 scatters the received blocks into its own film, which already holds rank 0's tiles.
    `force=True` (test aid) runs pack -> gather -> scatter even at world 1 and for rank 0's own block, so that a one-GPU box
    executes the process-group code path of the N>1 launch on the real RCCL."""

    def __init__(self, width, height, tile_size, rank, world, device, stage_host=False, force=False):
        import torch
        self.stage_host = stage_host  # gloo with device films (a test aid: gloo moves host tensors only)
        self.force = force
        self.rank, self.world, self.n_pixels = rank, world, width * height
        per_rank = [owned_pixels(width, height, tile_size[0], tile_size[1], r, world) for r in range(world)]
        self.counts = [len(p) for p in per_rank]
        self.block = max(self.counts) if self.counts else 0  # pixels per gathered block (uniform: a collective, not P2P)
        self.mine = torch.from_numpy(per_rank[rank]).to(device)
        self.device = device
        xdev = "cpu" if stage_host else device
        # rank 0's own block only carries data under `force`; otherwise it is the collective's (ignored) placeholder
        self.send = torch.zeros(self.block, 10, dtype=torch.float32, device=device)
        self.send_x = torch.zeros(self.block, 10, dtype=torch.float32, device=xdev) if stage_host else self.send
        if rank == 0:
            self.all = [torch.from_numpy(p).to(device) for p in per_rank]
            self.recv_x = [torch.zeros(self.block, 10, dtype=torch.float32, device=xdev) for _ in range(world)]

    def pack(self, film):
        """[count, 10] = Color 3 | Alpha 1 | Background 3 | WorldNormal 3 of the owned pixels, into the preallocated buffer."""
        n = self.counts[self.rank]
        buf = self.send[:n]
        buf[:, 0:3] = film["color"].view(-1, 3)[self.mine]
        buf[:, 3] = film["alpha"].view(-1)[self.mine]
        buf[:, 4:7] = film["background"].view(-1, 3)[self.mine]
        buf[:, 7:10] = film["normal"].view(-1, 3)[self.mine]
        return buf

    def gather(self, film, group=None):
        """One collective per frame: every rank's packed pixels go to rank 0, which writes them into ITS film (returned on
        rank 0, complete; None elsewhere)."""
        import torch.distributed as dist
        if self.world == 1 and not self.force:
            return film
        if self.rank != 0 or self.force:
            self.pack(film)
            if self.stage_host:
                self.send_x.copy_(self.send)
        dist.gather(self.send_x, gather_list=self.recv_x if self.rank == 0 else None, dst=0, group=group)
        if self.rank != 0:
            return None
        for r in range(0 if self.force else 1, self.world):
            n = self.counts[r]
            idx, part = self.all[r], self.recv_x[r][:n].to(self.device)
            film["color"].view(-1, 3)[idx] = part[:, 0:3]
            film["alpha"].view(-1)[idx] = part[:, 3]
            film["background"].view(-1, 3)[idx] = part[:, 4:7]
            film["normal"].view(-1, 3)[idx] = part[:, 7:10]
        return film
