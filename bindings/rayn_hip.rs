//! rayn_hip.rs — `extern "C"` binding of librayn_hip.so for rayn.  Copy to `src/hip.rs`, copy film_hip.rs to `src/film_hip.rs`
//! and apply bindings/rayn.patch (adds `mod hip; mod film_hip;`, the `describe()` methods of the scene types and three
//! `pub(crate)`s); link with `cargo rustc -- -L <dir of librayn_hip.so>` or a two-line build.rs.
//!
//! Mirrors include/rayn_hip.h field for field.  UNTESTED AS RUST: no Rust toolchain exists in the build environment
//!.  What IS tested (tests/test_bindings.py, every round): the `#[repr(C)]` field lists below — names, order
//! and scalar widths — are parsed and compared with the C header and with the sizes the compiled library reports through
//! `rayn_hip_sizeof`, so this file cannot drift from the ABI unnoticed.  `check_layout()` repeats the size check at start-up.
//!
//! Reference seam: `Film::render_frame_into` (src/film.rs:382-395); the flattened `World` (src/world.rs:7-13).
#![allow(dead_code)]
use std::os::raw::c_char;

pub const RAYN_MAX_HITABLES: usize = 16;
pub const RAYN_MAX_MATERIALS: usize = 16;
pub const RAYN_MAX_LIGHTS: usize = 16;
pub const RAYN_FIS_TABLE_SIZE: usize = 512; // FILTER_TABLE_SIZE, src/filter.rs:187

// rayn_hitable_kind / rayn_sdf_kind / rayn_material_kind / rayn_camera_kind of include/rayn_hip.h
pub const RAYN_HITABLE_SPHERE: u32 = 0;
pub const RAYN_HITABLE_TRACED_SDF: u32 = 1;
pub const RAYN_SDF_SPHERE: u32 = 0;
pub const RAYN_SDF_MANDELBOX: u32 = 1;
pub const RAYN_SDF_MANDELBULB: u32 = 2;
pub const RAYN_MAT_LAMBERTIAN: u32 = 0;
pub const RAYN_MAT_DIELECTRIC: u32 = 1;
pub const RAYN_MAT_SKY: u32 = 2;
pub const RAYN_MAT_EMISSIVE: u32 = 3;
pub const RAYN_CAM_PINHOLE: u32 = 0;
pub const RAYN_CAM_THIN_LENS: u32 = 1;
pub const RAYN_CAM_ORTHOGRAPHIC: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynVec3 {
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

/// Sphere<TR> (src/sphere.rs:7-87) | TracedSDF<S> (src/sdf.rs:12-102)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynHitable {
    pub kind: u32,     // 0 Sphere, 1 TracedSDF
    pub material: u32, // MaterialHandle(usize), src/material.rs:55-56
    pub center: RaynVec3,
    pub radius: f32,
    pub sdf_kind: u32,   // 0 sdfu::Sphere, 1 MandelBox (src/sdf.rs:104-188), 2 Mandelbulb (extension)
    pub iterations: u32, // MandelBox::new(iterations, ..)
    pub box_side: f32,   // BoxFold::new(side_length), src/sdf.rs:151
    pub min_radius: f32, // SphereFold::new(min_radius, fixed_radius), src/sdf.rs:172
    pub fixed_radius: f32,
    pub scale: f32,
    pub sdf_radius: f32,
    pub animated: u32, // closure transform_seq |t| center + center_vel * t (src/animation.rs:62-68)
    pub center_vel: RaynVec3,
    pub scale_vel: f32, // extension: MandelBox scale = |t| scale + scale_vel * t (0 = the reference's constant)
}

/// Lambertian | Dielectric (exponent = remapped roughness, src/material.rs:167-174) | Sky | Emissive
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynMaterial {
    pub kind: u32,
    pub a: RaynVec3,
    pub b: RaynVec3,
    pub exponent: f32,
}

/// SphereLight::new(pos, rad, emission), src/light.rs:26-34
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynLight {
    pub pos: RaynVec3,
    pub rad: f32,
    pub emission: RaynVec3,
    pub _pad: u32,
}

/// PinholeCamera | ThinLensCamera | OrthographicCamera (src/camera.rs:41-285)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynCamera {
    pub kind: u32,
    pub res_w: f32,
    pub res_h: f32,
    pub vfov_or_size: f32,
    pub origin: RaynVec3,
    pub at: RaynVec3,
    pub up: RaynVec3,
    pub aperture: f32,
    pub focus: RaynVec3,
    pub animated: u32, // bit 0 origin, 1 at, 2 up, 3 focus: the closure |t| base + vel * t
    pub origin_vel: RaynVec3,
    pub at_vel: RaynVec3,
    pub up_vel: RaynVec3,
    pub focus_vel: RaynVec3,
}

/// World (src/world.rs:7-13) + VolumeParams (src/volume.rs:1-5), scene order preserved
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RaynWorldDesc {
    pub n_hitables: u32,
    pub n_materials: u32,
    pub n_lights: u32,
    pub hitables: [RaynHitable; 16],
    pub materials: [RaynMaterial; 16],
    pub lights: [RaynLight; 16],
    pub camera: RaynCamera,
    pub has_scattering: u32,
    pub coeff_scattering: f32,
    pub has_extinction: u32,
    pub coeff_extinction: f32,
}

/// the arguments of Film::render_frame_into + the constants it reads
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynFrameParams {
    pub width: u32,
    pub height: u32,
    pub samples: u32,
    pub tile_w: u32,
    pub tile_h: u32,
    pub max_bounces: u32,
    pub volume_marches: u32,
    pub frame: u32,
    pub time_start: f32,
    pub time_end: f32,
    pub max_marches: u32,
    pub max_vis_marches: u32,
    pub sdf_detail_scale: f32,
    pub world_radius: f32,
    pub tile_first: u32,
    pub tile_step: u32,
}

/// counters + timings of the last render
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaynStats {
    pub paths: u64,
    pub segments: u64,
    pub shaded_slots: u64,
    pub tiles: u64,
    pub batches: u64,
    pub ms_total: f64,
    pub ms_raygen: f64,
    pub ms_extend: f64,
    pub ms_bin: f64,
    pub ms_shade: f64,
    pub ms_compact: f64,
    pub ms_resolve: f64,
    pub launches_extend: u64,
    pub launches_shade: u64,
    pub queue_bytes_bin: u64,
    pub ms_shadow: f64,
    pub ms_finish: f64,
    pub queue_bytes_compact: u64,
    pub shadow_jobs: u64,
}

pub enum RaynCtx {}

#[link(name = "rayn_hip")]
extern "C" {
    pub fn rayn_hip_create(device: i32, out: *mut *mut RaynCtx) -> i32;
    pub fn rayn_hip_create_multi(devices: *const i32, n_devices: i32, out: *mut *mut RaynCtx) -> i32;
    pub fn rayn_hip_device_count(ctx: *const RaynCtx) -> i32;
    pub fn rayn_hip_destroy(ctx: *mut RaynCtx);
    pub fn rayn_hip_last_error(ctx: *const RaynCtx) -> *const c_char;
    pub fn rayn_hip_upload_world(ctx: *mut RaynCtx, world: *const RaynWorldDesc) -> i32;
    pub fn rayn_hip_render_frame(
        ctx: *mut RaynCtx,
        p: *const RaynFrameParams,
        samples_1d: *const f32,
        samples_2d: *const f32,
        scramble: *const f32,
        fis_table: *const f32,
        out_color: *mut f32,
        out_alpha: *mut f32,
        out_background: *mut f32,
        out_normal: *mut f32,
    ) -> i32;
    pub fn rayn_hip_get_stats(ctx: *const RaynCtx, out: *mut RaynStats) -> i32;
    /// entry 0 .. rayn_hip_device_count() - 1 of a multi-device context: what THAT device did in the last frame
    pub fn rayn_hip_get_entry_stats(ctx: *const RaynCtx, entry: i32, out: *mut RaynStats) -> i32;
    pub fn rayn_hip_set_fma_policy(ctx: *mut RaynCtx, policy: i32) -> i32;
    pub fn rayn_hip_sizeof(which: i32) -> usize;
}

/// Start-up layout check against the library as compiled (indices: include/rayn_hip.h, rayn_hip_sizeof).
pub fn check_layout() -> Result<(), String> {
    use std::mem::size_of;
    let expect = [
        (0, size_of::<RaynWorldDesc>(), "RaynWorldDesc"),
        (1, size_of::<RaynFrameParams>(), "RaynFrameParams"),
        (2, size_of::<RaynStats>(), "RaynStats"),
        (3, size_of::<RaynHitable>(), "RaynHitable"),
        (4, size_of::<RaynMaterial>(), "RaynMaterial"),
        (5, size_of::<RaynLight>(), "RaynLight"),
        (6, size_of::<RaynCamera>(), "RaynCamera"),
    ];
    for (which, size, name) in expect.iter() {
        let lib = unsafe { rayn_hip_sizeof(*which) };
        if lib != *size {
            return Err(format!("{}: binding has {} bytes, librayn_hip.so has {}", name, size, lib));
        }
    }
    Ok(())
}

// ---- glue used by the `describe()` methods bindings/rayn.patch adds to rayn's scene types ---------------------------------

impl From<crate::math::Vec3> for RaynVec3 {
    fn from(v: crate::math::Vec3) -> Self {
        RaynVec3 { x: v.x, y: v.y, z: v.z }
    }
}

/// lane 0 of a wide value (`f32x4::as_ref` -> `[f32; 4]`, the idiom of src/animation.rs:39-41)
pub fn lane0(v: crate::math::f32x4) -> f32 {
    let lanes = v.as_ref();
    lanes[0]
}

fn moves(vel: crate::math::Vec3) -> bool {
    vel.x != 0.0 || vel.y != 0.0 || vel.z != 0.0
}

/// `(base, vel)` of `WSequenced::as_linear` into the descriptor: bit 0 origin, 1 at, 2 up, 3 focus of `animated`
impl RaynCamera {
    pub fn set_origin(&mut self, (base, vel): (crate::math::Vec3, crate::math::Vec3)) {
        self.origin = base.into();
        self.origin_vel = vel.into();
        self.animated |= (moves(vel) as u32) << 0;
    }
    pub fn set_at(&mut self, (base, vel): (crate::math::Vec3, crate::math::Vec3)) {
        self.at = base.into();
        self.at_vel = vel.into();
        self.animated |= (moves(vel) as u32) << 1;
    }
    pub fn set_up(&mut self, (base, vel): (crate::math::Vec3, crate::math::Vec3)) {
        self.up = base.into();
        self.up_vel = vel.into();
        self.animated |= (moves(vel) as u32) << 2;
    }
    pub fn set_focus(&mut self, (base, vel): (crate::math::Vec3, crate::math::Vec3)) {
        self.focus = base.into();
        self.focus_vel = vel.into();
        self.animated |= (moves(vel) as u32) << 3;
    }
}

/// Owning handle of a `rayn_ctx` (one GPU, or several: `rayn_hip_create_multi`).  Every failing call returns the library's
/// message (`rayn_hip_last_error`) as `Err(String)` - the hot path of the reference panics instead (src/film.rs:127,667).
pub struct Context {
    raw: *mut RaynCtx,
}

// the library serialises calls on one ctx itself (include/rayn_hip.h: "not re-entrant per ctx"); the handle is moved, not shared
unsafe impl Send for Context {}

impl Context {
    pub fn new(devices: &[i32]) -> Result<Self, String> {
        check_layout()?;
        let mut raw: *mut RaynCtx = std::ptr::null_mut();
        let rc = unsafe {
            if devices.len() == 1 {
                rayn_hip_create(devices[0], &mut raw)
            } else {
                rayn_hip_create_multi(devices.as_ptr(), devices.len() as i32, &mut raw)
            }
        };
        if rc != 0 || raw.is_null() {
            return Err(format!("rayn_hip_create{:?} failed with status {}", devices, rc));
        }
        Ok(Context { raw })
    }

    pub fn raw(&self) -> *mut RaynCtx {
        self.raw
    }

    /// status -> Result, with the library's text
    pub fn check(&self, rc: i32) -> Result<(), String> {
        if rc == 0 {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(rayn_hip_last_error(self.raw)) };
        Err(format!("librayn_hip status {}: {}", rc, msg.to_string_lossy()))
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { rayn_hip_destroy(self.raw) }
    }
}
