"""Multi-GPU film partition: one process per GPU, whole 16x16 tiles dealt round-robin to ranks
(tile k belongs to rank (k + k // world) % world — tiles are fully independent, src/film.rs:439-627, and their
cost is very uneven, so ranks are interleaved and the interleave rotates every `world` tiles), and ONE gather of each rank's owned pixels to
rank 0 at frame end (`dist.gather` = ncclGather on RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).  No other
communication touches the data path.  (Inside ONE process the C ABI does the same with peer copies: rayn_hip_create_multi.)

r5: what travels is the LIBRARY's packed planar film of a share (include/rayn_hip.h: Color 3N | Alpha N | Background 3N | WorldNormal 3N
for the share's N pixels, tile after tile, x-major inside a tile).  A rank other than 0 resolves its tiles STRAIGHT into its send
block (rayn_hip_render_frame_packed_device - there is no full-resolution film on it and no pack step), rank 0 renders its own tiles
into the film it returns and scatters every received block with ONE kernel launch (rayn_hip_unpack_share_device, the k_unpack_tiles
the multi-device context uses).  The torch fancy-index pack / scatter of r1-r4 is gone from the device path; an index form of the same
layout survives only for CPU tensors (the gloo tests, whose per-rank renderer is the CPU oracle)."""
import copy

import numpy as np


def tile_rects(width, height, tile_w, tile_h):
    """The reference's tile list, x-major, incl. the (res + res%tile)/tile quirk (src/film.rs:399-427)."""
    nx, ny = (width + width % tile_w) // tile_w, (height + height % tile_h) // tile_h
    return [(tx * tile_w, ty * tile_h, min(tx * tile_w + tile_w, width), min(ty * tile_h + tile_h, height))
            for tx in range(nx) for ty in range(ny)]


def owned_pixels(width, height, tile_w, tile_h, rank, world):
    """Film pixel indices (x + y*width) of the tiles rank owns, in the order of the share's packed film: tile after tile in
    reference tile order, x outer / y inner inside a tile (include/rayn_hip.h, rayn_share_pixels)."""
    idx = []
    for k, (x0, y0, x1, y1) in enumerate(tile_rects(width, height, tile_w, tile_h)):
        if (k + k // world) % world != rank or x1 <= x0 or y1 <= y0:
            continue
        xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), indexing="ij")
        idx.append((xs + ys * width).reshape(-1))
    return np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)


_PLANES = (("color", 3, 0), ("alpha", 1, 3), ("background", 3, 4), ("normal", 3, 7))  # (channel, floats per pixel, plane offset in pixels)


class FilmGather:
    """PREALLOCATED buffers for gathering a tile-partitioned film onto rank 0 (nothing is allocated inside a frame).  The exchange
    is ONE collective: `dist.gather` to rank 0 (ncclGather semantics on RCCL: every rank contributes a buffer of the same size, so
    the per-rank blocks are padded to the largest share - shares differ by < 4 %: the half-height last tile row; rank 0's own
    contribution is the buffer the collective's signature requires of the root and carries data only under `force`).

    Device films (ctx = the rank's rayn_amd.Context, params = its frame parameters):
        rank != 0 (and rank 0 under `force`):  gather.render(d_tables, film)  resolves the share into `send` (packed planar film)
        rank 0:                                gather.render(d_tables, film)  renders its tiles into `film`
        gather.gather(film)                    one dist.gather; rank 0 unpacks every block with one kernel launch each
    CPU films (ctx None - the gloo tests): gather.gather(film) packs / scatters the same layout with index ops.
    `force=True` (test aid) sends rank 0's own block through the collective too, also at world 1, so that a one-GPU box executes the
    process-group code path of the N>1 launch on the real RCCL."""

    def __init__(self, width, height, tile_size, rank, world, device, stage_host=False, force=False, ctx=None, params=None):
        import torch
        self.stage_host = stage_host  # gloo with device films (a test aid: gloo moves host tensors only)
        self.force = force
        self.rank, self.world, self.n_pixels = rank, world, width * height
        self.width, self.height, self.tile = width, height, (int(tile_size[0]), int(tile_size[1]))
        self.ctx, self.device = ctx, device
        self._unpack_stream = None  # rank 0, device films: the stream the unpack launches run on (created on first use)
        if ctx is not None:
            from .film import share_pixels
            assert params is not None and (params.width, params.height, params.tile_w, params.tile_h) == (width, height) + self.tile
            self.share_params = []
            for r in range(world):
                q = copy.copy(params)
                q.tile_first, q.tile_step = r, world
                self.share_params.append(q)
            self.counts = [share_pixels(q) for q in self.share_params]
            self.mine = None
        else:
            per_rank = [owned_pixels(width, height, self.tile[0], self.tile[1], r, world) for r in range(world)]
            self.counts = [len(p) for p in per_rank]
            self.mine = torch.from_numpy(per_rank[rank])
            if rank == 0:
                self.all = [torch.from_numpy(p) for p in per_rank]
        self.block = max(self.counts) if self.counts else 0  # pixels per gathered block (uniform: a collective, not P2P)
        self.sends_block = rank != 0 or force                # does this rank's block carry pixels?
        xdev = "cpu" if stage_host else device
        self.send = torch.zeros(self.block * 10, dtype=torch.float32, device=device)
        self.send_x = torch.zeros(self.block * 10, dtype=torch.float32, device=xdev) if stage_host else self.send
        if rank == 0:
            self.recv_all = torch.zeros(world, self.block * 10, dtype=torch.float32, device=xdev)
            self.recv_x = list(self.recv_all.unbind(0))  # the collective writes straight into the rows
            self.recv_dev = torch.zeros(world, self.block * 10, dtype=torch.float32, device=device) if stage_host else self.recv_all

    # ---- device path -------------------------------------------------------------------------------------------------
    def render(self, d_tables, film):
        """The rank's share: straight into the packed send block when that block travels, into `film` on rank 0."""
        if self.sends_block:
            self.ctx.render_packed(self.share_params[self.rank], d_tables, self.send)
        else:
            self.ctx.render_device(self.share_params[self.rank], d_tables, film)

    # ---- CPU-tensor path (tests): the same packed planar layout with index ops ---------------------------------------
    def pack(self, film):
        n = self.counts[self.rank]
        for ch, k, off in _PLANES:
            self.send[off * n:(off + k) * n] = film[ch].reshape(self.n_pixels, -1)[self.mine].reshape(-1)
        return self.send

    def _scatter_indexed(self, film, r, block):
        n, idx = self.counts[r], self.all[r]
        for ch, k, off in _PLANES:
            film[ch].view(self.n_pixels, -1)[idx] = block[off * n:(off + k) * n].view(n, k)

    def gather(self, film, group=None):
        """One collective per frame: every rank's packed block goes to rank 0, which writes the blocks into ITS film (returned on rank 0, None
        elsewhere).  A device film is returned STREAM-ORDERED on torch's current stream, not host-complete: work enqueued on that stream afterwards
        sees every pixel; a host read needs the usual synchronise."""
        import torch.distributed as dist
        if self.world == 1 and not self.force:
            return film
        if self.sends_block:
            if self.ctx is None:
                self.pack(film)
            if self.stage_host:
                self.send_x.copy_(self.send)
        dist.gather(self.send_x, gather_list=self.recv_x if self.rank == 0 else None, dst=0, group=group)
        if self.rank != 0:
            return None
        if self.stage_host:
            self.recv_dev.copy_(self.recv_all)
        ustream = None
        if self.ctx is not None:
            # dist.gather returned once the collective was ENQUEUED: ProcessGroupNCCL makes torch's CURRENT stream wait for it, not the host.  The unpack
            # kernels therefore run on a stream of this object that first waits for torch's current stream (= for the collective), and torch's current
            # stream waits for that stream after the last unpack: the returned film is STREAM-ORDERED on torch's current stream - complete for every
            # consumer on it (and for the next frame's gather, which ProcessGroupNCCL orders after the current stream, so `recv_all` is not overwritten
            # under a running unpack) - without a host wait and without relying on legacy null-stream semantics.
            import torch
            if self._unpack_stream is None:
                self._unpack_stream = torch.cuda.Stream(device=self.device)
            ustream = self._unpack_stream
            ustream.wait_stream(torch.cuda.current_stream())
        for r in range(0 if self.force else 1, self.world):
            if self.counts[r] == 0:
                continue
            if self.ctx is not None:
                self.ctx.unpack_share(self.share_params[r], self.recv_dev[r], film, stream=ustream.cuda_stream)
            else:
                self._scatter_indexed(film, r, self.recv_dev[r])
        if ustream is not None:
            import torch
            torch.cuda.current_stream().wait_stream(ustream)
        return film
