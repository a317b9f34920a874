//! film_hip.rs — `World -> RaynWorldDesc` flattening and the alternate `Film::render_frame_into` on librayn_hip.so.
//! Copy to rayn's `src/film_hip.rs` next to `src/hip.rs` (= bindings/rayn_hip.rs) and apply bindings/rayn.patch.
//! UNTESTED AS RUST (no toolchain here).  What IS tested every round (tests/test_bindings.py): the patch
//! applies to the reference, and every method / field / constant this file uses on a rayn type is defined by the patched
//! reference or by hip.rs; the arithmetic it replaces is tested through the same C ABI (ctypes, and compiled C++ in
//! tests/host_mirror.cpp).
//!
//! `Box<dyn Hitable>` etc. carry no type tag, so bindings/rayn.patch gives the four scene traits one method with a default,
//! `fn describe(&self) -> Option<Rayn...> { None }`, and overrides it for the closed set: `Sphere<TR>` with a constant or
//! `animation::Linear` centre (src/sphere.rs:7-21), `TracedSDF<MandelBox>` (src/sdf.rs:12-21,104-122; `BoxFold` / `SphereFold`
//! keep their constructor arguments), `Lambertian` / `Dielectric` / `Emissive` with constant parameters, `Sky`
//! (src/material.rs), `SphereLight` (src/light.rs:19-34) and the three cameras (src/camera.rs).  Anything else returns `None`
//! and `render_frame_into_hip` returns `Err` - src/main.rs (as patched) then takes the CPU path.
use crate::camera::CameraHandle;
use crate::film::{ChannelStorage, Film};
use crate::filter::{Filter, FilterImportanceSampler};
use crate::hip::*;
use crate::integrator::{Integrator, PathTracingIntegrator};
use crate::math::{Extent2u, Vec3};
use crate::sampler::Samples;
use crate::spectrum::Srgb;
use crate::world::World;

use generic_array::ArrayLength;
use rand::prelude::*;
use std::ops::Range;

impl World {
    /// src/world.rs:7-13, scene order preserved (HitableStore::add_hits folds in order, src/hitable.rs:170-210).
    /// `None` = some object is outside the closed set of the library.
    pub fn to_desc(&self, camera: CameraHandle) -> Option<RaynWorldDesc> {
        if self.hitables.len() > RAYN_MAX_HITABLES
            || self.materials.len() > RAYN_MAX_MATERIALS
            || self.lights.len() > RAYN_MAX_LIGHTS
        {
            return None;
        }
        let mut d = RaynWorldDesc {
            n_hitables: self.hitables.len() as u32,
            n_materials: self.materials.len() as u32,
            n_lights: self.lights.len() as u32,
            hitables: [RaynHitable::default(); RAYN_MAX_HITABLES],
            materials: [RaynMaterial::default(); RAYN_MAX_MATERIALS],
            lights: [RaynLight::default(); RAYN_MAX_LIGHTS],
            camera: self.cameras.get(camera).describe()?,
            has_scattering: 0,
            coeff_scattering: 0.0,
            has_extinction: 0,
            coeff_extinction: 0.0,
        };
        for (i, h) in self.hitables.iter().enumerate() {
            d.hitables[i] = h.describe()?;
        }
        for (i, m) in self.materials.iter().enumerate() {
            d.materials[i] = m.describe()?;
        }
        for (i, l) in self.lights.iter().enumerate() {
            d.lights[i] = l.describe()?;
        }
        if let Some(s) = self.volume_params.coeff_scattering {
            d.has_scattering = 1;
            d.coeff_scattering = s;
        }
        if let Some(t) = self.volume_params.coeff_extinction {
            d.has_extinction = 1;
            d.coeff_extinction = t;
        }
        Some(d)
    }
}

impl<N: ArrayLength<ChannelStorage>> Film<N> {
    /// `Film::render_frame_into` (src/film.rs:382-628) on the GPU(s) behind `ctx`.  Same arguments; the integrator is the
    /// concrete `PathTracingIntegrator` (the only one the reference has, src/integrator.rs:33-36).  The film channels this
    /// `Film` was created with are overwritten with the normalised frame exactly like `tile_finished` does
    /// (src/film.rs:660-691); channels it does not hold are rendered and dropped.
    #[allow(clippy::too_many_arguments)]
    pub fn render_frame_into_hip<F: Filter + Copy + Send>(
        &mut self,
        ctx: &Context,
        world: &World,
        camera: CameraHandle,
        integrator: &PathTracingIntegrator,
        filter: &F,
        tile_size: Extent2u,
        frame: usize,
        time_range: Range<f32>,
        samples: usize,
    ) -> Result<(), String> {
        let desc = world
            .to_desc(camera)
            .ok_or_else(|| String::from("scene outside the closed set of librayn_hip"))?;

        // the same host-side tables as the CPU path (src/film.rs:429-434, 460-461): plain inputs of the ABI
        let fis = FilterImportanceSampler::new(filter);
        let sets_1d = 1 + integrator.requested_1d_sample_sets();
        let sets_2d = 2 + integrator.requested_2d_sample_sets();
        let sample_sets = Samples::new_rd(4 * samples, sets_1d, sets_2d, frame as u64);
        let (w, h) = (self.res.w, self.res.h);
        let mut scramble = vec![0f32; (w * h) as usize];
        for y in 0..h {
            for x in 0..w {
                let mut rng = SmallRng::seed_from_u64((x + y * w) as u64);
                scramble[(x + y * w) as usize] = rng.gen();
            }
        }

        let p = RaynFrameParams {
            width: w,
            height: h,
            samples: samples as u32,
            tile_w: tile_size.w,
            tile_h: tile_size.h,
            max_bounces: integrator.max_bounces as u32,
            volume_marches: integrator.volume_marches as u32,
            frame: frame as u32,
            time_start: time_range.start,
            time_end: time_range.end,
            max_marches: crate::sdf::MAX_MARCHES,
            max_vis_marches: crate::sdf::MAX_VIS_MARCHES,
            sdf_detail_scale: crate::setup::SDF_DETAIL_SCALE,
            world_radius: crate::setup::WORLD_RADIUS,
            tile_first: 0,
            tile_step: 1,
        };

        // The library writes planar f32 channels; rayn's storages are Vec<Srgb> / Vec<f32> / Vec<Vec3> whose element layout
        // Rust does not promise (`Srgb(pub Vec3)` carries no #[repr]), so the frame lands in plain buffers and is copied
        // element by element: 40 B per pixel once per frame.
        let n = (w * h) as usize;
        let mut color = vec![0f32; 3 * n];
        let mut alpha = vec![0f32; n];
        let mut background = vec![0f32; 3 * n];
        let mut normal = vec![0f32; 3 * n];

        ctx.check(unsafe { rayn_hip_upload_world(ctx.raw(), &desc) })?;
        ctx.check(unsafe {
            rayn_hip_render_frame(
                ctx.raw(),
                &p,
                sample_sets.samples_1d.as_ptr(),
                sample_sets.samples_2d.as_ptr(),
                scramble.as_ptr(),
                fis.inverse_cdf.as_ptr(),
                color.as_mut_ptr(),
                alpha.as_mut_ptr(),
                background.as_mut_ptr(),
                normal.as_mut_ptr(),
            )
        })?;

        let mut channels = self.channels.lock().unwrap();
        for channel in channels.iter_mut() {
            match channel {
                ChannelStorage::Color(buf) => {
                    for (dst, src) in buf.iter_mut().zip(color.chunks_exact(3)) {
                        *dst = Srgb::new(src[0], src[1], src[2]);
                    }
                }
                ChannelStorage::Alpha(buf) => buf.copy_from_slice(&alpha),
                ChannelStorage::Background(buf) => {
                    for (dst, src) in buf.iter_mut().zip(background.chunks_exact(3)) {
                        *dst = Srgb::new(src[0], src[1], src[2]);
                    }
                }
                ChannelStorage::WorldNormal(buf) => {
                    for (dst, src) in buf.iter_mut().zip(normal.chunks_exact(3)) {
                        *dst = Vec3::new(src[0], src[1], src[2]);
                    }
                }
            }
        }
        drop(channels);
        self.progressive_epoch += 1;
        Ok(())
    }
}
