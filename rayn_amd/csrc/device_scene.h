// device_scene.h — POD scene as the kernels see it (uniform data, read through scalar loads).
// Built on the host from rayn_world_desc + rayn_frame_params (include/rayn_hip.h).
#pragma once
#include <stdint.h>

#include "../../include/rayn_hip.h"

namespace rayn {

struct f3 { float x, y, z; };

struct DHitable {
    uint32_t kind, material, sdf_kind, iterations;
    f3 center; float radius_sq;           // Sphere: f32x4::from(radius*radius), src/sphere.rs:31,52
    float box_l, min_rad_sq, fixed_rad_sq, scale; // MandelBox (src/sdf.rs:114-122,151-158,172-179)
    float sdf_radius; uint32_t fast_div; uint32_t animated; uint32_t _pad;
    f3 center_vel; float scale_vel; // EXTENSION: MandelBox scale(t0) = scale + scale_vel * t0 (rayn_hip.h); 0 = constant
};
struct DMaterial { uint32_t kind, receives_light; float exponent, _pad; f3 a; float _p1; f3 b; float _p2; };
struct DLight { f3 pos; float rad; f3 emission; float _pad; };
struct DCamera {
    uint32_t kind; float half_w, half_h, full_w, full_h, half_pixel_size, aperture, _pad;
    f3 origin; float _p0; f3 at; float _p1; f3 up; float _p2; f3 focus; float _p3;
    f3 origin_vel; uint32_t animated; f3 at_vel; float _p4; f3 up_vel; float _p5; f3 focus_vel; float _p6;
};

struct DScene {
    uint32_t n_hitables, n_materials, n_lights, n_sdf;
    DHitable h[RAYN_MAX_HITABLES];
    DMaterial m[RAYN_MAX_MATERIALS];
    DLight l[RAYN_MAX_LIGHTS];
    DCamera cam;
    uint32_t has_scatter, has_extinct; float rho_s, rho_t;
    uint32_t anim_spheres, _pad_a[3]; // any hitable (Sphere, or TracedSDF as an extension) with a time-sequenced origin
    // frame constants
    uint32_t width, height, spp, max_bounces, vm, max_marches, max_vis_marches, n1, n2;
    float time_start, time_range, detail_scale, t_max, ndc_x, ndc_y;
};

// One 16x16 (clamped) film tile of the current batch.  film_packed != 0: the resolve writes the tile's pixels at
// [film_base + local pixel] of a PLANAR film that holds only the owned tiles (multi-device peers, rayn_hip.hip render_multi:
// the buffer that crosses xGMI) instead of at (x, y) of a full-resolution film.
struct DTile { uint32_t x0, y0, ew, eh; uint32_t pool_base, n_paths; uint32_t film_base, film_packed; };

constexpr uint32_t INVALID = 0xFFFFFFFFu;
constexpr uint32_t OBJ_NONE = 0xFFu;
constexpr uint32_t TERM_NONE = 0xFFu;

} // namespace rayn
