//! film_hip.rs — `World -> RaynWorldDesc` flattening and the alternate `Film::render_frame_into` on librayn_hip.so.
//! Drop into rayn's `src/` next to `hip.rs` (bindings/rayn_hip.rs).  UNTESTED AS RUST (no toolchain here); the
//! arithmetic it replaces is tested through the same C ABI by tests/ (ctypes) and tests/host_mirror.cpp (compiled C++).
//!
//! `Box<dyn Hitable>` etc. carry no type tag, so each concrete type gains one method on a small trait
//! (`fn describe(&self) -> Option<RaynHitable>`): `Sphere<Vec3>` / `Sphere<F: Fn(f32) -> Vec3>` of the linear form
//! (src/sphere.rs:7-21), `TracedSDF<MandelBox>` (src/sdf.rs:12-21,104-122; `BoxFold` / `SphereFold` expose their
//! constructor arguments), `Lambertian` / `Dielectric<WSrgb, f32x4>` / `Sky` / `Emissive<WSrgb>` (src/material.rs),
//! `SphereLight` (src/light.rs:19-34) and the three cameras (src/camera.rs).  Anything outside the closed set returns
//! `None` and the caller keeps the CPU path.
use crate::hip::*;
use std::ffi::CStr;
use std::ops::Range;

impl World {
    /// src/world.rs:7-13, scene order preserved (HitableStore::add_hits folds in order, src/hitable.rs:170-210).
    pub fn to_desc(&self, camera: CameraHandle) -> Option<RaynWorldDesc> {
        let mut d: RaynWorldDesc = unsafe { std::mem::zeroed() };
        if self.hitables.len() > RAYN_MAX_HITABLES || self.lights.len() > RAYN_MAX_LIGHTS { return None; }
        d.n_hitables = self.hitables.len() as u32;
        for (i, h) in self.hitables.iter().enumerate() { d.hitables[i] = h.describe()?; }
        d.n_materials = self.materials.len() as u32; // needs MaterialStore::{len, iter}
        for (i, m) in self.materials.iter().enumerate() { d.materials[i] = m.describe()?; }
        d.n_lights = self.lights.len() as u32;
        for (i, l) in self.lights.iter().enumerate() { d.lights[i] = l.describe()?; }
        d.camera = self.cameras.get(camera).describe()?;
        if let Some(s) = self.volume_params.coeff_scattering { d.has_scattering = 1; d.coeff_scattering = s; }
        if let Some(t) = self.volume_params.coeff_extinction { d.has_extinction = 1; d.coeff_extinction = t; }
        Some(d)
    }
}

impl<N: ArrayLength<ChannelStorage> + ArrayLength<ChannelTileStorage>> Film<N> {
    /// Film::render_frame_into (src/film.rs:382-628) on the GPU(s) behind `ctx` (rayn_hip_create or rayn_hip_create_multi).
    pub fn render_frame_into_hip<F: Filter + Copy + Send>(
        &mut self, ctx: *mut RaynCtx, world: &World, camera: CameraHandle, integrator: &PathTracingIntegrator,
        filter: &F, tile_size: Extent2u, frame: usize, time_range: Range<f32>, samples: usize,
    ) -> Result<(), String> {
        let desc = world.to_desc(camera).ok_or("scene outside the closed set of librayn_hip")?;
        // the same host-side tables as the CPU path (src/film.rs:429-434, 460-461) - they are plain inputs of the ABI
        let fis = FilterImportanceSampler::new(filter); // [f32; 512]: make `inverse_cdf` pub(crate)
        let sets_1d = 1 + integrator.requested_1d_sample_sets();
        let sets_2d = 2 + integrator.requested_2d_sample_sets();
        let sample_sets = Samples::new_rd(4 * samples, sets_1d, sets_2d, frame as u64);
        let (w, h) = (self.res.w, self.res.h);
        let scramble: Vec<f32> = (0..h)
            .flat_map(|y| (0..w).map(move |x| SmallRng::seed_from_u64((x + y * w) as u64).gen::<f32>()))
            .collect();
        let p = RaynFrameParams {
            width: w, height: h, samples: samples as u32, tile_w: tile_size.w, tile_h: tile_size.h,
            max_bounces: integrator.max_bounces as u32, volume_marches: integrator.volume_marches as u32, frame: frame as u32,
            time_start: time_range.start, time_end: time_range.end,
            max_marches: 256, max_vis_marches: 100, // src/sdf.rs:9-10
            sdf_detail_scale: crate::setup::SDF_DETAIL_SCALE, world_radius: crate::setup::WORLD_RADIUS,
            tile_first: 0, tile_step: 1,
        };
        let mut ch = self.channels.lock().unwrap(); // Color: Vec<Srgb>, Alpha: Vec<f32>, Background: Vec<Srgb>, WorldNormal: Vec<Vec3>
        let (color, alpha, bg, normal) = channel_ptrs_mut(&mut ch); // Srgb / Vec3 are #[repr(C)] 3 x f32
        let rc = unsafe {
            rayn_hip_upload_world(ctx, &desc);
            rayn_hip_render_frame(ctx, &p, sample_sets.samples_1d.as_ptr(), sample_sets.samples_2d.as_ptr(),
                                  scramble.as_ptr(), fis.inverse_cdf.as_ptr(), color, alpha, bg, normal)
        };
        if rc != 0 {
            return Err(unsafe { CStr::from_ptr(rayn_hip_last_error(ctx)) }.to_string_lossy().into());
        }
        self.progressive_epoch += 1;
        Ok(())
    }
}
