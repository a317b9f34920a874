"""Multi-GPU film partition: one process per GPU, whole 16x16 tiles dealt round-robin to ranks
(tile k belongs to rank (k + k // world) % world — tiles are fully independent, src/film.rs:439-627, and their
cost is very uneven, so ranks are interleaved and the interleave rotates every `world` tiles), and ONE gather of each rank's owned pixels to
rank 0 at frame end (grouped point-to-point transfers: RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).  No other
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
    frame).  Every rank sends exactly its own pixel count (10 floats per pixel) with one point-to-point transfer; rank 0 posts
    all receives as one batch (a grouped ncclSend/ncclRecv on RCCL = the frame's single collective step) and scatters the
    pixels into its own film, which already holds rank 0's tiles."""

    def __init__(self, width, height, tile_size, rank, world, device, stage_host=False):
        import torch
        self.stage_host = stage_host  # gloo with device films (a test aid: gloo moves host tensors only)
        self.rank, self.world, self.n_pixels = rank, world, width * height
        per_rank = [owned_pixels(width, height, tile_size[0], tile_size[1], r, world) for r in range(world)]
        self.counts = [len(p) for p in per_rank]
        self.mine = torch.from_numpy(per_rank[rank]).to(device)
        self.device = device
        self.send = torch.empty(self.counts[rank], 10, dtype=torch.float32, device=device) if rank != 0 else None
        if rank == 0:
            self.all = [torch.from_numpy(p).to(device) for p in per_rank]
            self.recv = [None] + [torch.empty(self.counts[r], 10, dtype=torch.float32, device=device) for r in range(1, world)]

    def pack(self, film):
        """[count, 10] = Color 3 | Alpha 1 | Background 3 | WorldNormal 3 of the owned pixels, into the preallocated buffer."""
        buf = self.send
        buf[:, 0:3] = film["color"].view(-1, 3)[self.mine]
        buf[:, 3] = film["alpha"].view(-1)[self.mine]
        buf[:, 4:7] = film["background"].view(-1, 3)[self.mine]
        buf[:, 7:10] = film["normal"].view(-1, 3)[self.mine]
        return buf

    def gather(self, film, group=None):
        """One exchange per frame: ranks > 0 send their packed pixels to rank 0, which writes them into ITS film (returned on
        rank 0, complete; None elsewhere)."""
        import torch.distributed as dist
        if self.world == 1:
            return film
        if self.rank != 0:
            buf = self.pack(film)
            dist.send(buf.cpu() if self.stage_host else buf, dst=0, group=group)
            return None
        if self.stage_host:
            for r in range(1, self.world):
                host = self.recv[r].cpu()
                dist.recv(host, src=r, group=group)
                self.recv[r].copy_(host)
        else:
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, self.recv[r], r, group) for r in range(1, self.world)])
            for q in reqs:
                q.wait()
        for r in range(1, self.world):
            idx, part = self.all[r], self.recv[r]
            film["color"].view(-1, 3)[idx] = part[:, 0:3]
            film["alpha"].view(-1)[idx] = part[:, 3]
            film["background"].view(-1, 3)[idx] = part[:, 4:7]
            film["normal"].view(-1, 3)[idx] = part[:, 7:10]
        return film
