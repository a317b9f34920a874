"""Multi-GPU film partition: one process per GPU, whole 16x16 tiles dealt round-robin to ranks
(tile k belongs to rank (k + k // world) % world — tiles are fully independent, src/film.rs:439-627, and their
cost is very uneven, so ranks are interleaved and the interleave rotates every `world` tiles), and ONE gather of each rank's owned pixels to
rank 0 at frame end (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).  No other
collective touches the data path."""
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
    """Precomputed index plan for gathering a tile-partitioned film onto rank 0."""

    def __init__(self, width, height, tile_size, rank, world, device):
        import torch
        self.rank, self.world, self.n_pixels = rank, world, width * height
        per_rank = [owned_pixels(width, height, tile_size[0], tile_size[1], r, world) for r in range(world)]
        self.count_max = max(len(p) for p in per_rank)
        self.mine = torch.from_numpy(per_rank[rank]).to(device)
        self.all = [torch.from_numpy(p).to(device) for p in per_rank] if rank == 0 else None
        self.device = device

    def pack(self, film):
        """[count_max, 10] = Color 3 | Alpha 1 | Background 3 | WorldNormal 3 of the owned pixels."""
        import torch
        buf = torch.zeros(self.count_max, 10, dtype=torch.float32, device=self.device)
        n = self.mine.numel()
        buf[:n, 0:3] = film["color"].view(-1, 3)[self.mine]
        buf[:n, 3] = film["alpha"].view(-1)[self.mine]
        buf[:n, 4:7] = film["background"].view(-1, 3)[self.mine]
        buf[:n, 7:10] = film["normal"].view(-1, 3)[self.mine]
        return buf

    def gather(self, film, group=None):
        """One collective: every rank sends its packed pixels to rank 0, which scatters them into the
        full raster (returned on rank 0; None elsewhere)."""
        import torch
        import torch.distributed as dist
        buf = self.pack(film)
        if self.world == 1:
            parts = [buf]
        else:
            parts = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(buf, parts, dst=0, group=group)
        if self.rank != 0:
            return None
        out = {"color": torch.zeros(self.n_pixels, 3, dtype=torch.float32, device=self.device),
               "alpha": torch.zeros(self.n_pixels, dtype=torch.float32, device=self.device),
               "background": torch.zeros(self.n_pixels, 3, dtype=torch.float32, device=self.device),
               "normal": torch.zeros(self.n_pixels, 3, dtype=torch.float32, device=self.device)}
        for r, part in enumerate(parts):
            idx = self.all[r]
            n = idx.numel()
            out["color"][idx] = part[:n, 0:3]
            out["alpha"][idx] = part[:n, 3]
            out["background"][idx] = part[:n, 4:7]
            out["normal"][idx] = part[:n, 7:10]
        return out
