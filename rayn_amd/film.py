"""Host-side mirror of rayn's `Film` (src/film.rs:175-378) on top of the C ABI.

    film = Film([ChannelKind.Color, ChannelKind.Alpha, ChannelKind.Background, ChannelKind.WorldNormal], (w, h))
    film.render_frame_into(world, camera, integrator, filter, tile_size, frame, time_range, samples)
    film.save_to([ChannelKind.Color], "renders", "64_spp", False)

`render_frame_into` has the reference's signature (src/film.rs:382-395).  Device memory, the stream
and (for N>1 ranks) the process group come from PyTorch; the arithmetic is all in librayn_hip.so.
"""
import ctypes as C
import enum
import os

import numpy as np

from . import _abi, image
from ._lib import RaynHipError, lib
from .params import frame_params


class ChannelKind(enum.Enum):  # src/film.rs:103-120
    Color = 0
    Alpha = 1
    Background = 2
    WorldNormal = 3


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def build_tables(spp, max_bounces, volume_marches, frame, width, height, filter=None):
    """Samples::new_rd (src/sampler.rs:18-37), per-pixel scramble (src/film.rs:460-461) and
    FilterImportanceSampler::new (src/filter.rs:196-220) through the product's host builders."""
    L = lib()
    kind, radius = (0, 1.5) if filter is None else (filter.kind, filter.radius)
    p0, p1 = getattr(filter, "params", (0.0, 0.0))
    n1, n2 = L.rayn_sets_1d(max_bounces, volume_marches), L.rayn_sets_2d(max_bounces, volume_marches)
    s1, s2 = np.zeros(spp * n1, np.float32), np.zeros(spp * 2 * n2, np.float32)
    scr, fis = np.zeros(width * height, np.float32), np.zeros(_abi.FIS_TABLE_SIZE, np.float32)
    for rc in (L.rayn_build_rd_tables(spp, n1, n2, frame, _fp(s1), _fp(s2)), L.rayn_build_scramble(width, height, _fp(scr)),
               L.rayn_build_fis_table_ex(kind, radius, p0, p1, _fp(fis))):
        if rc != 0:
            raise RaynHipError(f"table builder failed: {rc}")
    return s1, s2, scr, fis


def share_pixels(params):
    """rayn_share_pixels: pixels of the tiles the share (params.tile_first, params.tile_step) owns; its packed film is 10 floats each."""
    return int(lib().rayn_share_pixels(C.byref(params)))


class Context:
    """One rayn_ctx (one GPU).  Fails loudly when the HIP library or a GPU is missing."""

    def __init__(self, device=0):
        """device: one GPU index, or a list of indices for a multi-device context (rayn_hip_create_multi; buffers passed
        to render_device then live on the first one)."""
        self._L = lib()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*device)
            rc = self._L.rayn_hip_create_multi(ids, len(device), C.byref(h))
            first = device[0] if len(device) else 0
        else:
            rc = self._L.rayn_hip_create(device, C.byref(h))
            first = device
        if rc != 0:
            raise RaynHipError(f"rayn_hip_create(device={device}) failed with {rc} (no usable GPU?)")
        self.h = h
        self.device = first

    def device_count(self):
        return self._L.rayn_hip_device_count(self.h)

    def _chk(self, rc):
        if rc != 0:
            raise RaynHipError(f"rayn_hip error {rc}: {self._L.rayn_hip_last_error(self.h).decode()}")

    def table_broadcasts(self):
        """multi-device context: peer copies of the sample tables made so far (rayn_hip_table_broadcasts)"""
        return int(self._L.rayn_hip_table_broadcasts(self.h))

    def last_error(self):
        return self._L.rayn_hip_last_error(self.h).decode()

    def upload_world(self, desc):
        self._chk(self._L.rayn_hip_upload_world(self.h, C.byref(desc)))

    def set_profiling(self, timing=True, count_evals=False):
        self._chk(self._L.rayn_hip_set_profiling(self.h, int(timing), int(count_evals)))

    def set_fma_policy(self, policy):
        """0 = unfused mul_add (reference default build, the default), 1 = fused (rayn built with +fma)."""
        self._chk(self._L.rayn_hip_set_fma_policy(self.h, int(policy)))

    def set_workers(self, n_workers, min_paths=1 << 22):
        self._chk(self._L.rayn_hip_set_workers(self.h, int(n_workers), int(min_paths)))

    def set_tile_subset(self, tiles=None):
        """Render only these tiles (reference tile order) until cleared with None / []."""
        arr = np.ascontiguousarray([] if tiles is None else tiles, dtype=np.uint32)
        self._chk(self._L.rayn_hip_set_tile_subset(self.h, arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(arr)))

    def set_trace_tile(self, tile_index):
        self._chk(self._L.rayn_hip_set_trace_tile(self.h, int(tile_index)))

    def trace(self):
        """Packet-order dump of the traced tile: dict of uint32 arrays (depth, obj, px, py, sample, valid)."""
        n = self._L.rayn_hip_get_trace(self.h, None, 0)
        buf = np.zeros((max(n, 0), 6), np.uint32)
        if n > 0:
            self._L.rayn_hip_get_trace(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return {k: buf[:, i].copy() for i, k in enumerate(("depth", "obj", "px", "py", "sample", "valid"))}

    def set_batch_paths(self, n):
        self._chk(self._L.rayn_hip_set_batch_paths(self.h, int(n)))

    def set_cold_bytes(self, n):
        """arena bytes of ALL workers together for the context's first frame (rayn_hip_set_cold_bytes, rayn_hip.h); 0 = full-size batches from the first frame"""
        self._chk(self._L.rayn_hip_set_cold_bytes(self.h, int(n)))

    def stats(self):
        s = _abi.Stats()
        self._chk(self._L.rayn_hip_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def entry_stats(self, entry):
        """What device `entry` of a multi-device context did in the last frame (rayn_hip_get_entry_stats)."""
        s = _abi.Stats()
        self._chk(self._L.rayn_hip_get_entry_stats(self.h, int(entry), C.byref(s)))
        return s.as_dict()

    def eval_counts(self):
        out = (C.c_uint64 * 3)()
        self._chk(self._L.rayn_hip_get_eval_counts(self.h, out))
        return {"extend": out[0], "shade_setup": out[1], "shadow": out[2]}

    def sdf_iterations(self):
        """Fold / orbit iterations run by the evaluations of eval_counts() (instrumented kernels, rayn_hip_get_sdf_iterations)."""
        out = (C.c_uint64 * 3)()
        self._chk(self._L.rayn_hip_get_sdf_iterations(self.h, out))
        return {"extend": out[0], "shade_setup": out[1], "shadow": out[2]}

    def elision_counts(self):
        """Instrumented kernels (rayn_hip_get_elision_counts): shaded slots with throughput exactly (0, 0, 0) whose NEE was elided, and the
        shadow segments they would have parked."""
        out = (C.c_uint64 * 3)()
        self._chk(self._L.rayn_hip_get_elision_counts(self.h, out))
        return {"zero_throughput_slots": out[0], "elided_shadow_jobs": out[1], "samples_out_of_bounds": out[2]}

    def stage_slots(self):
        """Instrumented Mandelbulb shadow-march kernel (rayn_hip_get_stage_slots): lane slots offered by the orbit / epilogue stage of k_shadow_bulb."""
        out = (C.c_uint64 * 2)()
        self._chk(self._L.rayn_hip_get_stage_slots(self.h, out))
        return {"shadow_orbit": out[0], "shadow_epilogue": out[1]}

    def render_host(self, params, tables, out=None):
        """rayn_hip_render_frame with host (numpy) buffers.  Returns the film dict."""
        s1, s2, scr, fis = tables
        n = params.width * params.height
        if out is None:
            out = {"color": np.zeros((n, 3), np.float32), "alpha": np.zeros(n, np.float32),
                   "background": np.zeros((n, 3), np.float32), "normal": np.zeros((n, 3), np.float32)}
        self._chk(self._L.rayn_hip_render_frame(self.h, C.byref(params), _fp(s1), _fp(s2), _fp(scr), _fp(fis), _fp(out["color"]),
                                                _fp(out["alpha"]), _fp(out["background"]), _fp(out["normal"])))
        h, w = params.height, params.width
        return {"color": out["color"].reshape(h, w, 3), "alpha": out["alpha"].reshape(h, w),
                "background": out["background"].reshape(h, w, 3), "normal": out["normal"].reshape(h, w, 3)}

    def render_device(self, params, d_tables, d_out, stream=None):
        """rayn_hip_render_frame_device: d_tables/d_out are torch CUDA tensors (kept resident in HBM)."""
        import torch
        ptr = lambda t: C.c_void_p(t.data_ptr())
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self._chk(self._L.rayn_hip_render_frame_device(self.h, C.byref(params), ptr(d_tables[0]), ptr(d_tables[1]), ptr(d_tables[2]),
                                                       ptr(d_tables[3]), ptr(d_out["color"]), ptr(d_out["alpha"]),
                                                       ptr(d_out["background"]), ptr(d_out["normal"]), C.c_void_p(s)))

    def render_packed(self, params, d_tables, d_packed, stream=None):
        """rayn_hip_render_frame_packed_device: the share (params.tile_first / tile_step) straight into its PACKED planar film
        `d_packed` (a float32 CUDA tensor of >= 10 * share_pixels(params) elements) - what a rank hands to the film gather."""
        import torch
        ptr = lambda t: C.c_void_p(t.data_ptr())
        assert d_packed.dtype == torch.float32 and d_packed.is_contiguous() and d_packed.numel() >= 10 * share_pixels(params)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self._chk(self._L.rayn_hip_render_frame_packed_device(self.h, C.byref(params), ptr(d_tables[0]), ptr(d_tables[1]), ptr(d_tables[2]),
                                                              ptr(d_tables[3]), ptr(d_packed), C.c_void_p(s)))

    def unpack_share(self, params, d_packed, d_out, stream=None):
        """rayn_hip_unpack_share_device: scatter the packed film of the share (params.tile_first / tile_step) into the full-resolution
        film `d_out` with one kernel launch on the stream (enqueued, not waited for)."""
        import torch
        ptr = lambda t: C.c_void_p(t.data_ptr())
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self._chk(self._L.rayn_hip_unpack_share_device(self.h, C.byref(params), ptr(d_packed), ptr(d_out["color"]), ptr(d_out["alpha"]),
                                                       ptr(d_out["background"]), ptr(d_out["normal"]), C.c_void_p(s)))

    def close(self):
        if self.h:
            self._L.rayn_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def alloc_device_film(width, height, device="cuda"):
    import torch
    n = width * height
    return {"color": torch.zeros(n, 3, dtype=torch.float32, device=device), "alpha": torch.zeros(n, dtype=torch.float32, device=device),
            "background": torch.zeros(n, 3, dtype=torch.float32, device=device), "normal": torch.zeros(n, 3, dtype=torch.float32, device=device)}


class Film:
    """Film::<U4>::new(channels, res) (src/film.rs:184-203)."""

    def __init__(self, channels, res, device=0):
        kinds = list(channels)
        if len(set(kinds)) != len(kinds):
            dup = next(k for k in kinds if kinds.count(k) > 1)
            raise ValueError(f"Attempted to create multiple {dup.name} channels")  # src/film.rs:188
        self.channel_kinds = kinds
        self.res = (int(res[0]), int(res[1]))
        self.ctx = Context(device)
        self.device = f"cuda:{device}"
        self.channels = None  # torch tensors after a render
        self.progressive_epoch = 0

    def render_frame_into(self, world, camera, integrator, filter, tile_size, frame, time_range, samples, tile_first=0, tile_step=1):
        """Film::render_frame_into (src/film.rs:382-628); the film is overwritten, not accumulated (:91)."""
        import torch
        w, h = self.res
        p = frame_params(w, h, samples, integrator.max_bounces, integrator.volume_marches, frame, time_range, tile_size, tile_first, tile_step)
        self.ctx.upload_world(world.to_desc(camera))
        tables = build_tables(4 * samples, integrator.max_bounces, integrator.volume_marches, frame, w, h, filter)
        with torch.cuda.device(self.device):
            d_tables = [torch.from_numpy(t).to(self.device) for t in tables]
            out = alloc_device_film(w, h, self.device)
            self.ctx.render_device(p, d_tables, out)
            torch.cuda.synchronize()
        self.channels = out
        self.progressive_epoch += 1
        return self.ctx.stats()

    def channel(self, kind):
        if kind not in self.channel_kinds:
            raise KeyError(f"Attempted to read {kind.name} channel but it didn't exist")
        key = {ChannelKind.Color: "color", ChannelKind.Alpha: "alpha", ChannelKind.Background: "background", ChannelKind.WorldNormal: "normal"}[kind]
        w, h = self.res
        t = self.channels[key].cpu().numpy()
        return t.reshape(h, w, 3) if t.ndim == 2 else t.reshape(h, w)

    def save_to(self, write_channels, output_folder, base_name, transparent_background=False):
        """Film::save_to (src/film.rs:205-378) - the post-process after the hot path, arm by arm; the reference's
        Err(String) cases raise ValueError with the same text."""
        os.makedirs(output_folder, exist_ok=True)
        have = lambda k: k in self.channel_kinds
        for kind in write_channels:
            if kind == ChannelKind.Color:
                if have(ChannelKind.Color) and have(ChannelKind.Alpha) and transparent_background:
                    img = image.color_image(self.channel(ChannelKind.Color), alpha=self.channel(ChannelKind.Alpha), transparent_background=True)
                elif have(ChannelKind.Color) and have(ChannelKind.Background) and not transparent_background:
                    img = image.color_image(self.channel(ChannelKind.Color), background=self.channel(ChannelKind.Background))
                elif have(ChannelKind.Color) and not have(ChannelKind.Background) and not transparent_background:
                    img = image.color_image(self.channel(ChannelKind.Color))
                else:
                    raise ValueError("Attempted to write Color channel with insufficient channels")
                image.save(os.path.join(output_folder, f"{base_name}_color.png"), img)
            elif kind == ChannelKind.Background:
                if not have(kind):
                    raise ValueError("Attempted to write Background channel but it didn't exist")
                image.save(os.path.join(output_folder, f"{base_name}_background.png"), image.background_image(self.channel(kind)))
            elif kind == ChannelKind.WorldNormal:
                if not have(kind):
                    raise ValueError("Attempted to write WorldNormal channel but it didn't exist")
                image.save(os.path.join(output_folder, f"{base_name}_normal.png"), image.normal_image(self.channel(kind)))
            elif kind == ChannelKind.Alpha:
                if not have(kind):
                    raise ValueError("Attempted to write Alpha channel but it didn't exist")
                image.save(os.path.join(output_folder, f"{base_name}_alpha.png"), image.alpha_image(self.channel(kind)))
