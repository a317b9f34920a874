"""Frame parameters = the arguments of Film::render_frame_into (src/film.rs:382-395) plus the
compile-time constants it reads (src/setup.rs:16-44, src/sdf.rs:9-10), as the C-ABI POD."""
from . import _abi
from . import setup as _setup


def frame_params(width, height, samples, max_bounces, volume_marches=_setup.VOLUME_MARCHES_PER_SAMPLE, frame=1,
                 time_range=None, tile_size=_setup.TILE_SIZE, tile_first=0, tile_step=1,
                 max_marches=_setup.MAX_MARCHES, max_vis_marches=_setup.MAX_VIS_MARCHES,
                 sdf_detail_scale=_setup.SDF_DETAIL_SCALE, world_radius=_setup.WORLD_RADIUS) -> _abi.FrameParams:
    import numpy as np
    if time_range is None:
        # src/main.rs:47-62: frame_start = frame * (1/frame_rate); frame_end = frame_start + shutter_speed
        f32 = np.float32
        inv = f32(1.0) / f32(_setup.FRAME_RATE)
        start = f32(frame) * inv
        time_range = (float(start), float(f32(start + f32(1.0) / f32(24.0))))
    p = _abi.FrameParams()
    p.width, p.height, p.samples = width, height, samples
    p.tile_w, p.tile_h = tile_size
    p.max_bounces, p.volume_marches, p.frame = max_bounces, volume_marches, frame
    p.time_start, p.time_end = time_range
    p.max_marches, p.max_vis_marches = max_marches, max_vis_marches
    p.sdf_detail_scale, p.world_radius = sdf_detail_scale, world_radius
    p.tile_first, p.tile_step = tile_first, tile_step
    return p
