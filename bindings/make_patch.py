"""Regenerates bindings/rayn.patch: the reference-side changes a rayn maintainer applies so that bindings/hip.rs +
bindings/film_hip.rs compile inside rayn's `src/` (N1).

The script copies the reference's `src/` + Cargo.toml into a scratch directory, applies the edits below as exact,
assert-checked text replacements (so a changed reference fails loudly instead of producing a stale patch) and writes the
unified diff.  It runs only in the build container (it reads /root/reference); the committed patch is what travels.
usage: python bindings/make_patch.py [reference_root=/root/reference] [out=bindings/rayn.patch]"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "rayn.patch")

EDITS = {}


GROUP = ["hip"]  # the group the following edit() calls belong to: "hip" (the backend binding), "dump" (RAYN_DUMP only), "both"


def edit(path, old, new, count=1):
    EDITS.setdefault(path, []).append((old, new, count, GROUP[0]))


# ---- src/main.rs: the three new modules + an opt-in switch next to the CPU call -------------------------------------
edit("src/main.rs", "mod film;\n", "mod film;\nmod film_hip;\n")
edit("src/main.rs", "mod hitable;\n", "mod hip;\nmod hitable;\n")
edit("src/main.rs", """    let (camera, world) = setup::setup();
""", """    let (camera, world) = setup::setup();

    // RAYN_HIP=0[,1,..]: render on those GPUs through librayn_hip.so (film_hip.rs); unset = the CPU path below.
    let hip_ctx = std::env::var("RAYN_HIP").ok().map(|ids| {
        let ids: Vec<i32> = ids.split(',').map(|s| s.trim().parse().expect("RAYN_HIP=<gpu>[,<gpu>..]")).collect();
        hip::Context::new(&ids).expect("librayn_hip")
    });
""")
edit("src/main.rs", """        film.render_frame_into(
            &world,
            camera,
            &integrator,
            &filter,
            Extent2u::new(16, 16),
            frame,
            frame_start..frame_end,
            crate::setup::SAMPLES,
        );
""", """        let on_gpu = match &hip_ctx {
            Some(ctx) => match film.render_frame_into_hip(
                ctx,
                &world,
                camera,
                &integrator,
                &filter,
                Extent2u::new(16, 16),
                frame,
                frame_start..frame_end,
                crate::setup::SAMPLES,
            ) {
                Ok(()) => true,
                Err(e) => {
                    println!("librayn_hip: {} - falling back to the CPU path", e);
                    false
                }
            },
            None => false,
        };
        if !on_gpu {
            film.render_frame_into(
                &world,
                camera,
                &integrator,
                &filter,
                Extent2u::new(16, 16),
                frame,
                frame_start..frame_end,
                crate::setup::SAMPLES,
            );
        }
""")

# ---- src/animation.rs: a sequenced value can say whether it is `base + vel * t`; the nameable linear closure --------
edit("src/animation.rs", """pub trait WSequenced<T>: Send + Sync {
    fn sample_at(&self, t: f32x4) -> T;
}
""", """pub trait WSequenced<T>: Send + Sync {
    fn sample_at(&self, t: f32x4) -> T;

    /// Closed-set description for the HIP backend (hip.rs): `Some((base, vel))` when the value is `base + vel * t`
    /// (`vel = 0` for a constant).  `None` - the default, e.g. an arbitrary closure - keeps the scene on the CPU path.
    fn as_linear(&self) -> Option<(Vec3, Vec3)> {
        None
    }

    /// `Some(c)` when the value is the constant scalar `c` (ThinLensCamera's aperture).
    fn as_constant(&self) -> Option<f32> {
        None
    }
}
""")
edit("src/animation.rs", "impl_wsequenced_for_sequenced!(f32 => f32x4, Vec2 => Wec2, Vec3 => Wec3);\n",
     """impl_wsequenced_for_sequenced!(Vec2 => Wec2);

// the `f32 => f32x4` and `Vec3 => Wec3` instances of the macro above, written out so that they can describe themselves
impl WSequenced<f32x4> for f32 {
    #[inline]
    fn sample_at(&self, t: f32x4) -> f32x4 {
        let ts = t.as_ref();
        f32x4::from([
            Sequenced::sample_at(self, ts[0]),
            Sequenced::sample_at(self, ts[1]),
            Sequenced::sample_at(self, ts[2]),
            Sequenced::sample_at(self, ts[3]),
        ])
    }

    fn as_constant(&self) -> Option<f32> {
        Some(*self)
    }
}

impl WSequenced<Wec3> for Vec3 {
    #[inline]
    fn sample_at(&self, t: f32x4) -> Wec3 {
        let ts = t.as_ref();
        Wec3::from([
            Sequenced::sample_at(self, ts[0]),
            Sequenced::sample_at(self, ts[1]),
            Sequenced::sample_at(self, ts[2]),
            Sequenced::sample_at(self, ts[3]),
        ])
    }

    fn as_linear(&self) -> Option<(Vec3, Vec3)> {
        Some((*self, Vec3::zero()))
    }
}

/// The closure `move |t| base + vel * t` as a nameable type: same values as the closure (evaluated, like every
/// `Fn(f32) -> Vec3`, at lane 0's time for all four lanes - see the impl below), but the HIP backend can read it.
#[derive(Clone, Copy)]
pub struct Linear {
    pub base: Vec3,
    pub vel: Vec3,
}

impl WSequenced<Wec3> for Linear {
    #[inline]
    fn sample_at(&self, t: f32x4) -> Wec3 {
        let ts = t.as_ref();
        let v = self.base + self.vel * ts[0];
        [v, v, v, v].into()
    }

    fn as_linear(&self) -> Option<(Vec3, Vec3)> {
        Some((self.base, self.vel))
    }
}
""")

# ---- the four traits gain `describe` with a default of None (= outside the closed set) ------------------------------
edit("src/hitable.rs", """        half_pixel_size_at: &dyn Fn(f32x4) -> f32x4,
    ) -> (MaterialHandle, WShadingPoint);
}
""", """        half_pixel_size_at: &dyn Fn(f32x4) -> f32x4,
    ) -> (MaterialHandle, WShadingPoint);

    /// POD description for the HIP backend (hip.rs); `None` = not in its closed set.
    fn describe(&self) -> Option<crate::hip::RaynHitable> {
        None
    }
}
""")
edit("src/material.rs", """        bump: &'bump Bump,
    ) -> &'bump mut dyn BSDF;
}
""", """        bump: &'bump Bump,
    ) -> &'bump mut dyn BSDF;

    /// POD description for the HIP backend (hip.rs); `None` = not in its closed set.
    fn describe(&self) -> Option<crate::hip::RaynMaterial> {
        None
    }
}
""")
edit("src/light.rs", """        max_distance: f32x4,
    ) -> (f32x4, f32x4);
}
""", """        max_distance: f32x4,
    ) -> (f32x4, f32x4);

    /// POD description for the HIP backend (hip.rs); `None` = not in its closed set.
    fn describe(&self) -> Option<crate::hip::RaynLight> {
        None
    }
}
""")
edit("src/camera.rs", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4;
}
""", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4;

    /// POD description for the HIP backend (hip.rs); `None` = not in its closed set.
    fn describe(&self) -> Option<crate::hip::RaynCamera> {
        None
    }
}
""")

# ---- src/sphere.rs ---------------------------------------------------------------------------------------------------
edit("src/sphere.rs", """        let t = f32x4::merge(take_t1, t1, t2);

        f32x4::merge(t1_valid | t2_valid, t, miss)
    }
""", """        let t = f32x4::merge(take_t1, t1, t2);

        f32x4::merge(t1_valid | t2_valid, t, miss)
    }

    fn describe(&self) -> Option<crate::hip::RaynHitable> {
        let (center, vel) = self.transform_seq.as_linear()?;
        let mut d = crate::hip::RaynHitable::default();
        d.kind = crate::hip::RAYN_HITABLE_SPHERE;
        d.material = self.material.0 as u32;
        d.center = center.into();
        d.radius = self.radius;
        d.animated = (vel.x != 0.0 || vel.y != 0.0 || vel.z != 0.0) as u32;
        d.center_vel = vel.into();
        Some(d)
    }
""")

# ---- src/sdf.rs: the SDFs keep their constructor arguments; TracedSDF<S> describes itself through S ------------------
edit("src/sdf.rs", "const MAX_MARCHES: u32 = 256;\nconst MAX_VIS_MARCHES: u32 = 100;\n",
     """pub(crate) const MAX_MARCHES: u32 = 256;
pub(crate) const MAX_VIS_MARCHES: u32 = 100;

/// What an SDF tells the HIP backend (hip.rs) about itself: fills the `sdf_*` fields of the descriptor and returns
/// true, or returns false (the default) when it is outside the closed set.  `TracedSDF<S>` needs `S: DescribeSdf`;
/// an SDF that should simply stay on the CPU path writes `impl DescribeSdf for MySdf {}`.
pub trait DescribeSdf {
    fn describe_sdf(&self, _into: &mut crate::hip::RaynHitable) -> bool {
        false
    }
}
""")
edit("src/sdf.rs", "impl<S: SDF<f32x4, Wec3> + Send + Sync> Hitable for TracedSDF<S> {\n",
     "impl<S: SDF<f32x4, Wec3> + DescribeSdf + Send + Sync> Hitable for TracedSDF<S> {\n")
edit("src/sdf.rs", """            WShadingPoint::new(hit, point, half_pixel_size, normal),
        )
    }
}
""", """            WShadingPoint::new(hit, point, half_pixel_size, normal),
        )
    }

    fn describe(&self) -> Option<crate::hip::RaynHitable> {
        let mut d = crate::hip::RaynHitable::default();
        d.kind = crate::hip::RAYN_HITABLE_TRACED_SDF;
        d.material = self.material.0 as u32;
        if self.sdf.describe_sdf(&mut d) {
            Some(d)
        } else {
            None
        }
    }
}
""")
edit("src/sdf.rs", """pub struct MandelBox {
    iterations: usize,
    scale: f32x4,
""", """pub struct MandelBox {
    iterations: usize,
    scale_arg: f32,
    scale: f32x4,
""")
edit("src/sdf.rs", """            sphere_fold,
            scale: scale.into(),
""", """            sphere_fold,
            scale_arg: scale,
            scale: scale.into(),
""")
edit("src/sdf.rs", """        let d = p.mag() / dr.abs();
        d
    }
}
""", """        let d = p.mag() / dr.abs();
        d
    }
}

impl DescribeSdf for MandelBox {
    fn describe_sdf(&self, into: &mut crate::hip::RaynHitable) -> bool {
        into.sdf_kind = crate::hip::RAYN_SDF_MANDELBOX;
        into.iterations = self.iterations as u32;
        into.box_side = self.box_fold.side_length;
        into.min_radius = self.sphere_fold.min_radius;
        into.fixed_radius = self.sphere_fold.fixed_radius;
        into.scale = self.scale_arg;
        true
    }
}
""")
edit("src/sdf.rs", """pub struct BoxFold {
    l: Wec3,
""", """pub struct BoxFold {
    side_length: f32,
    l: Wec3,
""")
edit("src/sdf.rs", """        BoxFold {
            l,
""", """        BoxFold {
            side_length,
            l,
""")
edit("src/sdf.rs", """pub struct SphereFold {
    min_rad_sq: f32x4,
""", """pub struct SphereFold {
    min_radius: f32,
    fixed_radius: f32,
    min_rad_sq: f32x4,
""")
edit("src/sdf.rs", """        Self {
            min_rad_sq,
            fixed_rad_sq,
        }
    }

    pub fn sphere_fold""", """        Self {
            min_radius,
            fixed_radius,
            min_rad_sq,
            fixed_rad_sq,
        }
    }

    pub fn sphere_fold""")

# ---- src/material.rs --------------------------------------------------------------------------------------------------
edit("src/material.rs", """    pub fn get(&self, handle: MaterialHandle) -> &dyn Material {
        self.0[handle.0].as_ref()
    }
}
""", """    pub fn get(&self, handle: MaterialHandle) -> &dyn Material {
        self.0[handle.0].as_ref()
    }

    pub fn len(&self) -> usize {
        self.0.len()
    }

    pub fn iter(&self) -> impl Iterator<Item = &dyn Material> + '_ {
        self.0.iter().map(|b| b.as_ref())
    }
}
""")
edit("src/material.rs", """pub trait WShadingParamGenerator<T> {
    fn gen(&self, intersection: &WShadingPoint) -> T;
}

impl<T, I: Into<T> + Copy> WShadingParamGenerator<T> for I {
    fn gen(&self, _intersection: &WShadingPoint) -> T {
        (*self).into()
    }
}
""", """pub trait WShadingParamGenerator<T> {
    fn gen(&self, intersection: &WShadingPoint) -> T;

    /// `Some(value)` when the parameter does not depend on the intersection (what the HIP backend can take).
    fn constant(&self) -> Option<T> {
        None
    }
}

impl<T, I: Into<T> + Copy> WShadingParamGenerator<T> for I {
    fn gen(&self, _intersection: &WShadingPoint) -> T {
        (*self).into()
    }

    fn constant(&self) -> Option<T> {
        Some((*self).into())
    }
}

/// lane 0 of a splatted colour (every lane of a constant parameter holds the same value)
fn lane0_rgb(c: WSrgb) -> crate::hip::RaynVec3 {
    crate::hip::RaynVec3 {
        x: crate::hip::lane0(c.x),
        y: crate::hip::lane0(c.y),
        z: crate::hip::lane0(c.z),
    }
}
""")
edit("src/material.rs", """        bump.alloc_with(|| LambertianBSDF {
            albedo: self.albedo_gen.gen(intersection),
        })
    }
}
""", """        bump.alloc_with(|| LambertianBSDF {
            albedo: self.albedo_gen.gen(intersection),
        })
    }

    fn describe(&self) -> Option<crate::hip::RaynMaterial> {
        let mut d = crate::hip::RaynMaterial::default();
        d.kind = crate::hip::RAYN_MAT_LAMBERTIAN;
        d.a = lane0_rgb(self.albedo_gen.constant()?);
        Some(d)
    }
}
""")
edit("src/material.rs", """        bump.alloc_with(|| DielectricBSDF {
            albedo: self.albedo_gen.gen(intersection),
            roughness: self.roughness_gen.gen(intersection),
        })
    }
}
""", """        bump.alloc_with(|| DielectricBSDF {
            albedo: self.albedo_gen.gen(intersection),
            roughness: self.roughness_gen.gen(intersection),
        })
    }

    fn describe(&self) -> Option<crate::hip::RaynMaterial> {
        let mut d = crate::hip::RaynMaterial::default();
        d.kind = crate::hip::RAYN_MAT_DIELECTRIC;
        d.a = lane0_rgb(self.albedo_gen.constant()?);
        // the REMAPPED roughness (Dielectric::new_remap above), i.e. the Phong exponent DielectricBSDF works with
        d.exponent = crate::hip::lane0(self.roughness_gen.constant()?);
        Some(d)
    }
}
""")
edit("src/material.rs", """        bump.alloc_with(|| SkyBSDF {
            top: WSrgb::splat(self.top),
            bottom: WSrgb::splat(self.bottom),
        })
    }
}
""", """        bump.alloc_with(|| SkyBSDF {
            top: WSrgb::splat(self.top),
            bottom: WSrgb::splat(self.bottom),
        })
    }

    fn describe(&self) -> Option<crate::hip::RaynMaterial> {
        let mut d = crate::hip::RaynMaterial::default();
        d.kind = crate::hip::RAYN_MAT_SKY;
        d.a = (*self.top).into();
        d.b = (*self.bottom).into();
        Some(d)
    }
}
""")
edit("src/material.rs", """            inner: LambertianBSDF {
                albedo: WSrgb::new_splat(0.5, 0.5, 0.5),
            },
        })
    }
}
""", """            inner: LambertianBSDF {
                albedo: WSrgb::new_splat(0.5, 0.5, 0.5),
            },
        })
    }

    fn describe(&self) -> Option<crate::hip::RaynMaterial> {
        let mut d = crate::hip::RaynMaterial::default();
        d.kind = crate::hip::RAYN_MAT_EMISSIVE;
        d.a = lane0_rgb(self.emission_gen.constant()?);
        Some(d)
    }
}
""")

# ---- src/light.rs: SphereLight keeps its scalar constructor arguments -----------------------------------------------
edit("src/light.rs", """pub struct SphereLight {
    pos: Wec3,
""", """pub struct SphereLight {
    desc: crate::hip::RaynLight,
    pos: Wec3,
""")
edit("src/light.rs", """        Self {
            pos: Wec3::splat(pos),
""", """        Self {
            desc: crate::hip::RaynLight {
                pos: pos.into(),
                rad,
                emission: (*emission).into(),
                _pad: 0,
            },
            pos: Wec3::splat(pos),
""")
edit("src/light.rs", """impl Light for SphereLight {
""", """impl Light for SphereLight {
    fn describe(&self) -> Option<crate::hip::RaynLight> {
        Some(self.desc)
    }

""")

# ---- src/camera.rs: the cameras keep (resolution, vfov | vertical_size) ---------------------------------------------
edit("src/camera.rs", """pub struct PinholeCamera<O, A, U> {
    half_size: Wec2,
""", """pub struct PinholeCamera<O, A, U> {
    resolution: Vec2,
    vfov: f32,
    half_size: Wec2,
""")
edit("src/camera.rs", """        PinholeCamera {
            half_size: Wec2::splat(Vec2::new(half_width, half_height)),
""", """        PinholeCamera {
            resolution,
            vfov,
            half_size: Wec2::splat(Vec2::new(half_width, half_height)),
""")
edit("src/camera.rs", """pub struct ThinLensCamera<A, O, LA, U, F> {
    half_size: Wec2,
""", """pub struct ThinLensCamera<A, O, LA, U, F> {
    resolution: Vec2,
    vfov: f32,
    half_size: Wec2,
""")
edit("src/camera.rs", """        ThinLensCamera {
            half_size: Wec2::splat(Vec2::new(half_width, half_height)),
""", """        ThinLensCamera {
            resolution,
            vfov,
            half_size: Wec2::splat(Vec2::new(half_width, half_height)),
""")
edit("src/camera.rs", """pub struct OrthographicCamera<O, A, U> {
    half_size: Wec2,
""", """pub struct OrthographicCamera<O, A, U> {
    resolution: Vec2,
    vertical_size: f32,
    half_size: Wec2,
""")
edit("src/camera.rs", """        Self {
            half_size: Wec2::splat(size / 2.0),
""", """        Self {
            resolution,
            vertical_size,
            half_size: Wec2::splat(size / 2.0),
""")
edit("src/camera.rs", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4 {
        self.half_pixel_size * t
    }
}
#[derive(Clone, Copy)]
pub struct ThinLensCamera""", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4 {
        self.half_pixel_size * t
    }

    fn describe(&self) -> Option<crate::hip::RaynCamera> {
        let mut d = crate::hip::RaynCamera::default();
        d.kind = crate::hip::RAYN_CAM_PINHOLE;
        d.res_w = self.resolution.x;
        d.res_h = self.resolution.y;
        d.vfov_or_size = self.vfov;
        d.set_origin(self.origin.as_linear()?);
        d.set_at(self.at.as_linear()?);
        d.set_up(self.up.as_linear()?);
        Some(d)
    }
}
#[derive(Clone, Copy)]
pub struct ThinLensCamera""")
edit("src/camera.rs", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4 {
        self.half_pixel_size * t
    }
}

#[derive(Clone, Copy)]
pub struct OrthographicCamera""", """    fn half_pixel_size_at(&self, t: f32x4) -> f32x4 {
        self.half_pixel_size * t
    }

    fn describe(&self) -> Option<crate::hip::RaynCamera> {
        let mut d = crate::hip::RaynCamera::default();
        d.kind = crate::hip::RAYN_CAM_THIN_LENS;
        d.res_w = self.resolution.x;
        d.res_h = self.resolution.y;
        d.vfov_or_size = self.vfov;
        d.aperture = self.aperture.as_constant()?;
        d.set_origin(self.origin.as_linear()?);
        d.set_at(self.at.as_linear()?);
        d.set_up(self.up.as_linear()?);
        d.set_focus(self.focus.as_linear()?);
        Some(d)
    }
}

#[derive(Clone, Copy)]
pub struct OrthographicCamera""")
edit("src/camera.rs", """    fn half_pixel_size_at(&self, _t: f32x4) -> f32x4 {
        self.half_pixel_size
    }
}
""", """    fn half_pixel_size_at(&self, _t: f32x4) -> f32x4 {
        self.half_pixel_size
    }

    fn describe(&self) -> Option<crate::hip::RaynCamera> {
        let mut d = crate::hip::RaynCamera::default();
        d.kind = crate::hip::RAYN_CAM_ORTHOGRAPHIC;
        d.res_w = self.resolution.x;
        d.res_h = self.resolution.y;
        d.vfov_or_size = self.vertical_size;
        d.set_origin(self.origin.as_linear()?);
        d.set_at(self.at.as_linear()?);
        d.set_up(self.up.as_linear()?);
        Some(d)
    }
}
""")

# ---- src/film.rs, src/filter.rs: what film_hip.rs (another module) has to reach -------------------------------------
edit("src/film.rs", """    channels: Mutex<GenericArray<ChannelStorage, N>>,
    progressive_epoch: usize,
    this_epoch_tiles_finished: AtomicUsize,
    res: Extent2u,
}
""", """    pub(crate) channels: Mutex<GenericArray<ChannelStorage, N>>,
    pub(crate) progressive_epoch: usize,
    this_epoch_tiles_finished: AtomicUsize,
    pub(crate) res: Extent2u,
}
""")
GROUP[0] = "both"
edit("src/filter.rs", """pub struct FilterImportanceSampler {
    inverse_cdf: [f32; FILTER_TABLE_SIZE],
}
""", """pub struct FilterImportanceSampler {
    pub(crate) inverse_cdf: [f32; FILTER_TABLE_SIZE],
}
""")

GROUP[0] = "dump"
# ---- RAYN_DUMP=<dir>[,<tile>]: the observable state of one render in tools/rayn_dump.py's format (N2) -------
# src/dump.rs is a NEW file carried by the patch; src/film.rs writes through it, src/hitable.rs exposes the bin lengths.
NEW_FILES = {}
NEW_FILES["src/dump.rs"] = """//! RAYN_DUMP=<dir>[,<tile>] - write what one `Film::render_frame_into` consumed and produced as raw little-endian arrays
//! (the format of librayn_hip's tools/rayn_dump.py, which then diffs two such directories array by array):
//!   samples_1d.f32 samples_2d.f32   Samples::new_rd tables                               src/sampler.rs:18-37
//!   scramble.f32                    SmallRng::seed_from_u64(x + y*w).gen::<f32>()        src/film.rs:460-461
//!   fis.f32                         FilterImportanceSampler::inverse_cdf (512)           src/filter.rs:187-220
//!   color / alpha / background / normal .f32   the film after tile_finished, index x + y*w (y = 0 is the bottom row)
//!   trace.u32                       tile <tile>'s per-depth packet lanes in HitStore::process_hits order, 6 u32 per lane:
//!                                   depth, object id, tile x, tile y, sample, valid      src/hitable.rs:94-134
//!   manifest.json                   the parameters the comparer checks first
//! Unset = no effect.  The dump happens inside the frame: do not time a frame that dumps.
use std::io::Write;

pub struct DumpCfg {
    pub dir: String,
    pub tile: usize,
}

/// `RAYN_DUMP=<dir>[,<tile>]` (tile defaults to 1, like tools/rayn_dump.py's --tile)
pub fn config() -> Option<DumpCfg> {
    let v = std::env::var("RAYN_DUMP").ok()?;
    let mut parts = v.splitn(2, ',');
    let dir = parts.next()?.to_string();
    let tile = parts.next().and_then(|t| t.trim().parse().ok()).unwrap_or(1);
    std::fs::create_dir_all(&dir).ok()?;
    Some(DumpCfg { dir, tile })
}

pub fn write_f32(dir: &str, name: &str, v: &[f32]) {
    let mut bytes = Vec::with_capacity(v.len() * 4);
    for x in v {
        bytes.extend_from_slice(&x.to_bits().to_le_bytes());
    }
    std::fs::write(format!("{}/{}.f32", dir, name), bytes).unwrap();
}

pub fn write_u32(dir: &str, name: &str, v: &[u32]) {
    let mut bytes = Vec::with_capacity(v.len() * 4);
    for x in v {
        bytes.extend_from_slice(&x.to_le_bytes());
    }
    std::fs::write(format!("{}/{}.u32", dir, name), bytes).unwrap();
}

#[allow(clippy::too_many_arguments)]
pub fn write_manifest(
    cfg: &DumpCfg,
    width: u32,
    height: u32,
    samples: usize,
    max_bounces: usize,
    volume_marches: usize,
    frame: usize,
    time_start: f32,
    time_end: f32,
    tile_w: u32,
    tile_h: u32,
) {
    let mut f = std::fs::File::create(format!("{}/manifest.json", cfg.dir)).unwrap();
    write!(
        f,
        "{{\\"scene\\": \\"rayn setup::setup()\\", \\"width\\": {}, \\"height\\": {}, \\"SAMPLES\\": {}, \\"spp\\": {}, \\"max_bounces\\": {}, \\"volume_marches\\": {}, \\"frame\\": {}, \\"time_range\\": [{:e}, {:e}], \\"tile\\": [{}, {}], \\"trace_tile\\": {}, \\"backend\\": \\"rayn\\"}}\\n",
        width,
        height,
        samples,
        4 * samples,
        max_bounces,
        volume_marches,
        frame,
        time_start,
        time_end,
        tile_w,
        tile_h,
        cfg.tile
    )
    .unwrap();
}
"""
edit("src/main.rs", "mod camera;\n", "mod camera;\nmod dump;\n")
edit("src/hitable.rs", """    pub fn reset(&mut self) {
        for hit in self.hits.iter_mut() {
            hit.clear();
        }
    }
""", """    pub fn reset(&mut self) {
        for hit in self.hits.iter_mut() {
            hit.clear();
        }
    }

    /// Length of every object's bin (after `process_hits`: padded to a multiple of 4 = whole packets), in object order.
    pub fn bin_lens<'a>(&'a self) -> impl Iterator<Item = usize> + 'a {
        self.hits.iter().map(|hits| hits.len())
    }
""")
edit("src/film.rs", """        let width = self.res.w;

        self.integrate_tiles(tiles, samples * 4, |tile| {
""", """        let width = self.res.w;

        // RAYN_DUMP=<dir>[,<tile>] (src/dump.rs): the tables this frame reads, one tile's packets, the film it leaves
        let dump_cfg = crate::dump::config();
        if let Some(cfg) = &dump_cfg {
            crate::dump::write_f32(&cfg.dir, "samples_1d", &sample_sets.samples_1d);
            crate::dump::write_f32(&cfg.dir, "samples_2d", &sample_sets.samples_2d);
            crate::dump::write_f32(&cfg.dir, "fis", &fis.inverse_cdf);
            let mut scramble = Vec::with_capacity((self.res.w * self.res.h) as usize);
            for y in 0..self.res.h {
                for x in 0..self.res.w {
                    let mut rng = SmallRng::seed_from_u64((x + y * width) as u64);
                    scramble.push(rng.gen::<f32>());
                }
            }
            crate::dump::write_f32(&cfg.dir, "scramble", &scramble);
            crate::dump::write_manifest(
                cfg,
                self.res.w,
                self.res.h,
                samples,
                crate::setup::MAX_INDIRECT_BOUNCES,
                VOLUME_MARCHES_PER_SAMPLE,
                frame,
                time_range.start,
                time_range.end,
                tile_size.w,
                tile_size.h,
            );
        }
        let dump_ref = dump_cfg.as_ref();

        self.integrate_tiles(tiles, samples * 4, |tile| {
            let trace_this = dump_ref.map_or(false, |cfg| cfg.tile == tile._index);
            let mut trace: Vec<u32> = Vec::new();
""")
edit("src/film.rs", """                hit_store.process_hits(&world.hitables, &mut wintersections, &half_pixel_size_at);
""", """                hit_store.process_hits(&world.hitables, &mut wintersections, &half_pixel_size_at);

                if trace_this {
                    // wintersections is object-major, every object's bin padded to whole packets (src/hitable.rs:94-134)
                    let mut k = 0;
                    for (obj_id, bin_len) in hit_store.bin_lens().enumerate() {
                        for _ in 0..bin_len / 4 {
                            let (_mat, wsp) = &wintersections[k];
                            k += 1;
                            for lane in 0..4 {
                                trace.extend_from_slice(&[
                                    depth as u32,
                                    obj_id as u32,
                                    wsp.ray.tile_coord[lane].x,
                                    wsp.ray.tile_coord[lane].y,
                                    wsp.ray.sample[lane] as u32,
                                    wsp.ray.valid[lane] as u32,
                                ]);
                            }
                        }
                    }
                }
""")
edit("src/film.rs", """                    spawned_wrays.push(wray);
                }
                spawned_rays.clear();
            }
        });
    }
""", """                    spawned_wrays.push(wray);
                }
                spawned_rays.clear();
            }

            if trace_this {
                if let Some(cfg) = dump_ref {
                    crate::dump::write_u32(&cfg.dir, "trace", &trace);
                }
            }
        });

        if let Some(cfg) = &dump_cfg {
            self.dump_channels(&cfg.dir);
        }
    }

    /// The film after `tile_finished` (src/film.rs:660-691) as flat f32 arrays, index x + y*w, y = 0 the bottom row.
    fn dump_channels(&self, dir: &str) {
        let channels = self.channels.lock().unwrap();
        for channel in channels.iter() {
            match channel {
                ChannelStorage::Color(buf) => {
                    let flat: Vec<f32> = buf.iter().flat_map(|s| vec![s.x, s.y, s.z]).collect();
                    crate::dump::write_f32(dir, "color", &flat);
                }
                ChannelStorage::Alpha(buf) => crate::dump::write_f32(dir, "alpha", buf),
                ChannelStorage::Background(buf) => {
                    let flat: Vec<f32> = buf.iter().flat_map(|s| vec![s.x, s.y, s.z]).collect();
                    crate::dump::write_f32(dir, "background", &flat);
                }
                ChannelStorage::WorldNormal(buf) => {
                    let flat: Vec<f32> = buf.iter().flat_map(|s| vec![s.x, s.y, s.z]).collect();
                    crate::dump::write_f32(dir, "normal", &flat);
                }
            }
        }
    }
""")


def generate(groups, out):
    """unified diff of the reference's src/ with the edits of `groups` applied"""
    tmp = tempfile.mkdtemp(prefix="rayn_patch_")
    try:
        for side in ("a", "b"):
            os.makedirs(os.path.join(tmp, side))
            shutil.copytree(os.path.join(REF, "src"), os.path.join(tmp, side, "src"))
        n_files = 0
        for path, edits in EDITS.items():
            f = os.path.join(tmp, "b", path)
            text = open(f).read()
            touched = False
            for old, new, count, group in edits:
                if group not in groups and group != "both":
                    continue
                assert text.count(old) == count, f"{path}: expected {count} occurrence(s) of {old!r}, found {text.count(old)}"
                text = text.replace(old, new)
                touched = True
            n_files += touched
            open(f, "w").write(text)
        if "dump" in groups:
            for path, text in NEW_FILES.items():
                assert not os.path.exists(os.path.join(tmp, "a", path)), path
                open(os.path.join(tmp, "b", path), "w").write(text)
                n_files += 1
        r = subprocess.run(["diff", "-ruN", "a/src", "b/src"], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 1, r.stderr
        # drop diff's timestamps: the patch should not change from run to run
        lines = []
        for ln in r.stdout.splitlines(keepends=True):
            if ln.startswith("diff -ruN "):
                continue
            if ln.startswith(("--- a/", "+++ b/")):
                ln = ln.split("\t")[0] + "\n"
                if ln.startswith("--- a/") and ln[6:].strip() in NEW_FILES:
                    ln = "--- /dev/null\n"  # a file the patch creates (git apply wants the null source)
            lines.append(ln)
        open(out, "w").write("".join(lines))
        print(f"wrote {out}: {len(lines)} lines, {n_files} files")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    # rayn.patch: everything (the HIP backend binding + the dump hooks).  rayn_dump.patch: ONLY the RAYN_DUMP hooks - what
    # tools/pin_against_rayn.sh applies to pin the CPU oracle against an unmodified rayn: no GPU, no librayn_hip.so to link.
    generate(("hip", "dump"), OUT)
    generate(("dump",), os.path.join(os.path.dirname(OUT), "rayn_dump.patch") if len(sys.argv) <= 2 else OUT.replace(".patch", "_dump.patch"))


if __name__ == "__main__":
    main()
