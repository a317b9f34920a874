/* rayn_hip.h — C ABI of the MI355X-native replacement for rayn's per-sample integrator hot path.
 *
 * The reference (fu5ha/rayn, a single Rust binary crate) has no FFI: the replaceable seam is
 *   Film::render_frame_into(&mut self, world:&World, camera:CameraHandle, integrator:&I, filter:&F,
 *                           tile_size:Extent2u, frame:usize, time_range:Range<f32>, samples:usize)
 *   (src/film.rs:382-395), i.e. everything between building the tile list (src/film.rs:397-427) and
 *   tile_finished (src/film.rs:660-691), together with the trait surface that closure drives:
 *   Hitable (src/hitable.rs:8-18), Material/BSDF (src/material.rs:11-38), Light (src/light.rs:5-17),
 *   Camera (src/camera.rs:5-19), Integrator (src/integrator.rs:13-30), Filter (src/filter.rs:7-10).
 * Trait objects cannot cross to a GPU, so the ABI takes the same surface as a CLOSED SET of POD
 * descriptors (every concrete type the reference ships) in scene order — order is semantic
 * (HitableStore::add_hits folds in order, src/hitable.rs:170-210; HitStore::process_hits emits
 * packets object-major, src/hitable.rs:94-134).
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return 0 on success or a
 * negative rayn_status; rayn_hip_last_error() gives the text.  Where the reference panics
 * (src/film.rs:127,160,197,667; src/material.rs:431) this ABI returns an error code instead.
 */
#ifndef RAYN_HIP_H
#define RAYN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAYN_MAX_HITABLES 16
#define RAYN_MAX_MATERIALS 16
#define RAYN_MAX_LIGHTS 16
#define RAYN_FIS_TABLE_SIZE 512 /* FILTER_TABLE_SIZE, src/filter.rs:187 */
#define RAYN_FILM_FLOATS_PER_PIXEL 10 /* Color 3 + Alpha 1 + Background 3 + WorldNormal 3 (src/film.rs:103-120) */

typedef enum {
    RAYN_OK = 0,
    RAYN_ERR_INVALID_ARG = -1,
    RAYN_ERR_NO_DEVICE = -2,
    RAYN_ERR_HIP = -3,
    RAYN_ERR_NO_WORLD = -4,
    RAYN_ERR_OOM = -5
} rayn_status;

typedef struct { float x, y, z; } rayn_vec3;

/* ---- Hitable (src/hitable.rs:8-18) -------------------------------------------------------- */
typedef enum {
    RAYN_HITABLE_SPHERE = 0,    /* Sphere<TR>, src/sphere.rs:7-87 (TR = constant Vec3) */
    RAYN_HITABLE_TRACED_SDF = 1 /* TracedSDF<S>, src/sdf.rs:12-102 */
} rayn_hitable_kind;

typedef enum {
    RAYN_SDF_SPHERE = 0,   /* sdfu::Sphere: |p| - r (used by BASELINE config 1) */
    RAYN_SDF_MANDELBOX = 1, /* MandelBox, src/sdf.rs:104-188 */
    /* EXTENSION, not in the reference (which has no Mandelbulb): power-8 Mandelbulb distance
     * estimator in the trigonometry-free polynomial form (I. Quilez), 'iterations' orbit steps with bailout
     * |w|^2 > 256, d = 0.25*ln(m)*sqrt(m)/dz.  Exists because BASELINE.json names the workload "Mandelbulb". */
    RAYN_SDF_MANDELBULB = 2
} rayn_sdf_kind;

typedef struct {
    uint32_t kind;     /* rayn_hitable_kind */
    uint32_t material; /* MaterialHandle(usize), src/material.rs:55-56 */
    /* RAYN_HITABLE_SPHERE: Sphere::new(transform_seq, radius, material), src/sphere.rs:14-20 */
    rayn_vec3 center;
    float radius;
    /* RAYN_HITABLE_TRACED_SDF: TracedSDF::new(sdf, material), src/sdf.rs:17-21 */
    uint32_t sdf_kind;   /* rayn_sdf_kind */
    uint32_t iterations; /* MandelBox::new(iterations, ..), src/sdf.rs:114 */
    float box_side;      /* BoxFold::new(side_length), src/sdf.rs:151 */
    float min_radius;    /* SphereFold::new(min_radius, fixed_radius), src/sdf.rs:172 */
    float fixed_radius;
    float scale;         /* MandelBox scale, src/sdf.rs:114 */
    float sdf_radius;    /* RAYN_SDF_SPHERE radius */
    /* RAYN_HITABLE_SPHERE with a closure transform_seq (TR: Fn(f32) -> Vec3, src/sphere.rs:7, src/animation.rs:62-68):
     * animated != 0 selects the linear closure |t| center + center_vel * t, evaluated — like the reference — at the ray
     * time of LANE 0 of the packet that calls hit / occluded / get_shading_info.
     * EXTENSION for RAYN_HITABLE_TRACED_SDF (the reference's TracedSDF has no transform and ignores time,
     * src/sdf.rs:12-25): center / animated / center_vel are honoured the same way - the SDF is evaluated in
     * the frame translated by that origin (hit: ray origin - origin; occluded: both ends - origin;
     * get_shading_info: normal estimated at point - origin, the shading point stays in world space).  This
     * is what gives BASELINE config 5 its "animated fractal with time-sampled motion blur".  A zero,
     * non-animated center (all the reference can express) is an exact no-op. */
    uint32_t animated;
    rayn_vec3 center_vel;
    /* EXTENSION for RAYN_SDF_MANDELBOX (scene S3: "time-varying fold/scale params"; the reference's SDFs
     * ignore time, src/sdf.rs:25): the scale as the closure |t| scale + scale_vel * t, evaluated - like every closure-sequenced
     * parameter (src/animation.rs:62-68) - at the ray time of lane 0 of the packet that calls hit / occluded /
     * get_shading_info.  0 (what a zeroed struct holds; the field used to be padding) is the reference's constant scale. */
    float scale_vel;
} rayn_hitable;

/* ---- Material / BSDF (src/material.rs:11-38) ---------------------------------------------- */
typedef enum {
    RAYN_MAT_LAMBERTIAN = 0, /* src/material.rs:85-142 */
    RAYN_MAT_DIELECTRIC = 1, /* src/material.rs:144-257; 'exponent' is the REMAPPED roughness */
    RAYN_MAT_SKY = 2,        /* src/material.rs:394-449 */
    RAYN_MAT_EMISSIVE = 3    /* src/material.rs:451-520 (inner Lambertian 0.5 is never sampled) */
} rayn_material_kind;

typedef struct {
    uint32_t kind;  /* rayn_material_kind */
    rayn_vec3 a;    /* albedo | albedo | sky top | emission */
    rayn_vec3 b;    /* -      | -      | sky bottom | - */
    float exponent; /* Dielectric: 1 + (1-roughness)^4 * 300, Dielectric::new_remap src/material.rs:167-174 */
} rayn_material;

/* ---- Light (src/light.rs:5-17): SphereLight::new(pos, rad, emission), src/light.rs:26-34 -- */
typedef struct {
    rayn_vec3 pos;
    float rad;
    rayn_vec3 emission;
    uint32_t _pad;
} rayn_light;

/* ---- Camera (src/camera.rs:5-19) ---------------------------------------------------------- */
typedef enum {
    RAYN_CAM_PINHOLE = 0,  /* src/camera.rs:41-119 */
    RAYN_CAM_THIN_LENS = 1, /* src/camera.rs:120-213 */
    RAYN_CAM_ORTHOGRAPHIC = 2 /* src/camera.rs:215-285 */
} rayn_camera_kind;

typedef struct {
    uint32_t kind;      /* rayn_camera_kind */
    float res_w, res_h; /* 'resolution: Vec2' argument of ::new */
    float vfov_or_size; /* vfov in degrees (pinhole, thin lens) | vertical_size (orthographic) */
    rayn_vec3 origin, at, up;
    float aperture;     /* thin lens */
    rayn_vec3 focus;    /* thin lens */
    /* Time-sequenced parameters (src/animation.rs): the reference lets origin/at/up/focus be closures
     * Fn(f32) -> Vec3.  Closures cannot cross the ABI; the closed set offers the linear closure
     * |t| base + vel * t per parameter (bit 0 origin, 1 at, 2 up, 3 focus of 'animated').  As in the
     * reference (src/animation.rs:62-68) a closure is evaluated at the time of LANE 0 of the ray-gen
     * packet (sample 4*floor(s/4) of the pixel) for all four lanes. */
    uint32_t animated;
    rayn_vec3 origin_vel, at_vel, up_vel, focus_vel;
} rayn_camera;

/* ---- World (src/world.rs:7-13) + VolumeParams (src/volume.rs:1-5) ------------------------- */
typedef struct {
    uint32_t n_hitables, n_materials, n_lights;
    rayn_hitable hitables[RAYN_MAX_HITABLES];    /* HitableStore, scene order */
    rayn_material materials[RAYN_MAX_MATERIALS]; /* MaterialStore */
    rayn_light lights[RAYN_MAX_LIGHTS];          /* Vec<Box<dyn Light>> */
    rayn_camera camera;                          /* the CameraHandle passed to render_frame_into */
    uint32_t has_scattering; /* coeff_scattering: Option<f32> */
    float coeff_scattering;
    uint32_t has_extinction; /* coeff_extinction: Option<f32> */
    float coeff_extinction;
} rayn_world_desc;

/* ---- arguments of Film::render_frame_into + the constants it reads ------------------------ */
typedef struct {
    uint32_t width, height;   /* Film.res, src/film.rs:180 */
    uint32_t samples;         /* 'samples' (spp = 4*samples), src/film.rs:391,434,439; closed-set limit: <= 4096 (16384 spp) */
    uint32_t tile_w, tile_h;  /* tile_size, src/main.rs:69 */
    uint32_t max_bounces;     /* PathTracingIntegrator.max_bounces, src/integrator.rs:34; closed-set limit: <= 120 */
    uint32_t volume_marches;  /* VOLUME_MARCHES_PER_SAMPLE (>= 2: samples_1d[3],[4] are indexed), src/setup.rs:25 */
    uint32_t frame;           /* seeds the sample tables, src/film.rs:434 */
    float time_start, time_end; /* time_range, src/main.rs:61-62 */
    uint32_t max_marches;     /* MAX_MARCHES = 256, src/sdf.rs:9 */
    uint32_t max_vis_marches; /* MAX_VIS_MARCHES = 100, src/sdf.rs:10 */
    float sdf_detail_scale;   /* SDF_DETAIL_SCALE, src/setup.rs:37 */
    float world_radius;       /* WORLD_RADIUS, src/setup.rs:33 */
    /* multi-GPU film partition (no reference counterpart; tiles are independent, src/film.rs:439-627):
     * this call renders the tiles k (reference tile order) with (k + k / tile_step) % tile_step == tile_first: every run of
     * tile_step consecutive tiles holds each owner once, and the assignment rotates from run to run so that no owner is
     * locked to a fixed lattice of image rows (measured: the plain k % 8 lattice left one of 8 ranks 7 % slower). */
    uint32_t tile_first, tile_step;
} rayn_frame_params;

/* counters + timings of the last render (device work only) */
typedef struct {
    uint64_t paths;          /* camera paths started */
    uint64_t segments;       /* valid rays extended (one per path per depth reached) */
    uint64_t shaded_slots;   /* packet lanes shaded incl. padding lanes */
    uint64_t tiles;
    uint64_t batches;
    double ms_total;         /* HIP-event time of the whole render on the ctx stream */
    double ms_raygen, ms_extend, ms_bin, ms_shade, ms_compact, ms_resolve; /* ms_shade = k_shade_setup */
    uint64_t launches_extend, launches_shade;
    uint64_t queue_bytes_bin;     /* algorithmic HBM bytes of the bin stage */
    double ms_shadow, ms_finish; /* k_shadow, k_shade_finish */
    uint64_t queue_bytes_compact; /* algorithmic HBM bytes of the repack stage */
    uint64_t shadow_jobs;         /* shadow segments marched by k_shadow (29 algorithmic bytes each: 4 ref + 24 segment in, 1 visibility out) */
} rayn_stats;

typedef struct rayn_ctx rayn_ctx;

/* Film::new (src/film.rs:184-203) + device selection: one ctx = one GPU ... */
int rayn_hip_create(int device, rayn_ctx** out);
/* ... or several ("rayn_hip_create(device_ids[], n)", the call being internally multi-GPU): one
 * context over n_devices GPUs of this process, replacing rayon's tile tasks (src/film.rs:630-691) with one renderer per GPU.
 * Every other entry point takes it like a single-device ctx; ALL device pointers passed to rayn_hip_render_frame_device
 * then live on devices[0].  Per frame the tiles of the call's share are dealt to the devices in rotation (the j-th owned
 * tile goes to device (j + j / n) % n), scene and tables are replicated, every device renders its tiles with its own
 * streams, and each device other than devices[0] sends the pixels of its tiles to devices[0] with ONE peer copy over xGMI
 * (hipMemcpyPeerAsync of 10 floats per pixel) - the only data that crosses devices.  A device id may repeat (the entries
 * then share that GPU and split its memory budget; used by the single-GPU tests).  rayn_hip_set_trace_tile returns
 * RAYN_ERR_INVALID_ARG on a multi-device ctx. */
int rayn_hip_create_multi(const int* devices, int n_devices, rayn_ctx** out);
int rayn_hip_device_count(const rayn_ctx* ctx); /* entries of the context (1 for rayn_hip_create) */
/* Multi-device context: the sample tables / scramble / filter table are copied to every peer when a frame's (four table pointers, width, height, samples,
 * max_bounces, volume_marches, frame) differ from the last broadcast to that peer - not every frame (r6).  A host that REWRITES its tables in place under
 * unchanged parameters calls rayn_hip_upload_world again (it forgets the broadcast) or passes other buffers.  Returns the peer copies of the tables made so
 * far (one per peer and broadcast; diagnostics / tests). */
uint64_t rayn_hip_table_broadcasts(const rayn_ctx* ctx);
void rayn_hip_destroy(rayn_ctx* ctx);
const char* rayn_hip_last_error(const rayn_ctx* ctx);

/* setup::setup() result (src/setup.rs:46-170) flattened; replaces passing &World. */
int rayn_hip_upload_world(rayn_ctx* ctx, const rayn_world_desc* world);

/* Film::render_frame_into (src/film.rs:382-628) incl. tile_finished's normalisation
 * (src/film.rs:82-98,660-691).  HOST pointers.  Tables are the ones Samples::new_rd
 * (src/sampler.rs:18-37), the per-pixel SmallRng scramble (src/film.rs:460-461) and
 * FilterImportanceSampler::new (src/filter.rs:187-220) produce:
 *   samples_1d: spp*sets_1d floats, samples_2d: 2*spp*sets_2d floats, scramble: width*height,
 *   fis_table: 512.  Outputs are full-resolution, bottom-up rows like the reference's film
 *   (the flip happens only in save_to, src/film.rs:236): color/background/normal 3 floats per
 *   pixel interleaved, alpha 1.  Pixels of tiles this call does not own are left untouched. */
int rayn_hip_render_frame(rayn_ctx* ctx, const rayn_frame_params* p,
                          const float* samples_1d, const float* samples_2d,
                          const float* scramble, const float* fis_table,
                          float* out_color, float* out_alpha, float* out_background,
                          float* out_normal);

/* Same, but every pointer is a DEVICE pointer on the ctx's GPU and the work is enqueued on
 * 'hip_stream' (a hipStream_t; NULL = the ctx's own stream), after everything already queued there.
 * The call is BLOCKING: it returns when the frame is complete (it waits once, at the end, to read the
 * frame statistics back; queue sizes stay on the device and the depth loop of a frame with <= 8 bounces never synchronises -
 * deeper ones read the queue size back every 4th depth from depth 8 on, to stop enqueueing depths for a batch whose paths have
 * all terminated).
 * This is the entry the bench and the multi-GPU paths use. */
int rayn_hip_render_frame_device(rayn_ctx* ctx, const rayn_frame_params* p,
                                 const float* d_samples_1d, const float* d_samples_2d,
                                 const float* d_scramble, const float* d_fis_table,
                                 float* d_out_color, float* d_out_alpha, float* d_out_background,
                                 float* d_out_normal, void* hip_stream);

int rayn_hip_get_stats(const rayn_ctx* ctx, rayn_stats* out);
/* entry 0 .. rayn_hip_device_count() - 1 of a multi-device context: what THAT device did in the last frame (its own tiles, segments,
 * batches; ms_total = HIP events around its share on its own stream, table broadcast and film copy excluded) - rayn_hip_get_stats holds
 * the sums and, in ms_total, the whole multi-device frame on devices[0]'s clock.  A single-device ctx has the one entry 0. */
int rayn_hip_get_entry_stats(const rayn_ctx* ctx, int entry, rayn_stats* out);

/* ---- the film gather of a multi-PROCESS launch (one process per GPU; no reference counterpart: tiles are independent,
 * src/film.rs:439-627, and the reference's tile_finished copies each finished tile into the one film, src/film.rs:660-691) ----
 * A rank that owns the share (tile_first, tile_step) renders it straight into a PACKED PLANAR film of its own pixels - no
 * full-resolution film exists on that rank and nothing is packed afterwards: Color 3N | Alpha N | Background 3N | WorldNormal 3N
 * floats for the N = rayn_share_pixels(p) pixels of its tiles, tile after tile in ascending reference tile order, pixel-major (x outer,
 * y inner) inside a tile.  That buffer is what crosses xGMI (ONE ncclGather / peer copy); the receiving rank scatters it into its
 * full-resolution film with rayn_hip_unpack_share_device (p carrying the SENDER's tile_first / tile_step): one kernel launch,
 * enqueued on hip_stream and not waited for; the share's tile list is uploaded the first time a share is seen and cached in the ctx for its
 * lifetime (one list of <= 32 B per tile per distinct (resolution, tile size, tile_first, tile_step): a fixed partition costs N - 1 lists).
 * It is the same packed layout and the same two kernels rayn_hip_create_multi uses between the devices of one process. */
uint64_t rayn_share_pixels(const rayn_frame_params* p);
int rayn_hip_render_frame_packed_device(rayn_ctx* ctx, const rayn_frame_params* p,
                                        const float* d_samples_1d, const float* d_samples_2d,
                                        const float* d_scramble, const float* d_fis_table,
                                        float* d_packed /* 10 * rayn_share_pixels(p) floats */, void* hip_stream);
int rayn_hip_unpack_share_device(rayn_ctx* ctx, const rayn_frame_params* p, const float* d_packed,
                                 float* d_out_color, float* d_out_alpha, float* d_out_background,
                                 float* d_out_normal, void* hip_stream);

/* ---- host-side table builders (the a1/a3/a4 rows of SURVEY.md section 8) ------------------ */
/* 1 + requested_1d_sample_sets(), 2 + requested_2d_sample_sets() (src/film.rs:431-432,
 * src/integrator.rs:39-45). */
uint32_t rayn_sets_1d(uint32_t max_bounces, uint32_t volume_marches);
uint32_t rayn_sets_2d(uint32_t max_bounces, uint32_t volume_marches);
/* Samples::new_rd(spp, sets_1d, sets_2d, frame), src/sampler.rs:18-37. */
int rayn_build_rd_tables(uint32_t spp, uint32_t sets_1d, uint32_t sets_2d, uint64_t frame,
                         float* samples_1d, float* samples_2d);
/* SmallRng::seed_from_u64(x + y*width).gen::<f32>() for every pixel, src/film.rs:460-461. */
int rayn_build_scramble(uint32_t width, uint32_t height, float* scramble);
/* FilterImportanceSampler::new(&F::new(..)), src/filter.rs:187-220, for the reference's four Filter
 * implementations (src/filter.rs:12-49 BlackmanHarris, :51-108 MitchellNetravali(radius, b, c),
 * :110-140 Box, :142-185 LanczosSinc(radius, tau)).  rayn_build_fis_table takes the two
 * parameter-free kinds; _ex takes all four (param0/param1 = b/c for Mitchell, tau/unused for
 * Lanczos).  The sampler assumes a filter without negative lobes (src/filter.rs:194-195); like
 * the reference the builder does not check that. */
enum { RAYN_FILTER_BLACKMAN_HARRIS = 0, RAYN_FILTER_BOX = 1, RAYN_FILTER_MITCHELL = 2, RAYN_FILTER_LANCZOS = 3 };
int rayn_build_fis_table(uint32_t filter_kind, float radius, float* table512);
int rayn_build_fis_table_ex(uint32_t filter_kind, float radius, float param0, float param1,
                            float* table512);
/* number of tiles render_frame_into builds, incl. its under-coverage quirk (src/film.rs:399-404). */
uint32_t rayn_tile_count(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h);

/* ---- diagnostics (no reference counterpart) ------------------------------------------------ */
/* timing: bracket every kernel launch with HIP events and fill rayn_stats.ms_*; count_evals: run
 * the instrumented kernel variants that count SDF distance evaluations (roofline accounting). */
int rayn_hip_set_profiling(rayn_ctx* ctx, int timing, int count_evals);
/* out[0] k_extend (closest-hit marches), out[1] k_shade_setup (normal estimation), out[2] k_shadow (NEE visibility) */
int rayn_hip_get_eval_counts(const rayn_ctx* ctx, uint64_t out[3]);
/* the fold / orbit ITERATIONS those evaluations ran, same three kernels (MandelBox::dist always runs `iterations` folds, src/sdf.rs:125-141;
 * the Mandelbulb extension stops at its bailout; a sphere SDF has none): what an evaluation of the scene's own SDF costs is
 * flop_per_iteration * iterations / evaluations + the epilogue - bench.py prices the roofline with it instead of a fixed figure */
int rayn_hip_get_sdf_iterations(const rayn_ctx* ctx, uint64_t out[3]);
/* r6, same instrumented kernels: out[0] = shaded slots whose throughput was exactly (0, 0, 0) and whose NEE was therefore elided - every Le / surface / volume
 * NEE term of src/integrator.rs:70,91-92,128-129 is multiplied by that zero, so their shadow rays are never marched (k_shade_setup, exact: see the
 * kernel) - out[1] = the shadow segments those slots would have parked for TracedSDF::occluded (src/sdf.rs:25-57), out[2] = NEE samples of such
 * slots that were NOT elided because their contribution is not provably finite (then inf * 0 / NaN must keep its bits: tested and marched as ever) */
int rayn_hip_get_elision_counts(const rayn_ctx* ctx, uint64_t out[3]);
/* r6, same instrumented kernels, single-Mandelbulb scenes only (rayn_amd/csrc/march_bulb.h; zero otherwise): the shadow-march kernel written for that SDF runs an
 * evaluation in two stages - orbit steps, then distance + one step of TracedSDF::occluded (src/sdf.rs:25-57) - each at its own occupancy.
 * out[0] / out[1] = lane slots (64 x wave executions) the orbit / epilogue stage of k_shadow_bulb offered; the lane slots USED are
 * rayn_hip_get_sdf_iterations()[2] (orbit steps) and rayn_hip_get_eval_counts()[2] (epilogues). */
int rayn_hip_get_stage_slots(const rayn_ctx* ctx, uint64_t out[2]);
/* A frame's tiles are dealt to up to two workers (host thread + HIP stream + own device memory each) so that one
 * worker's HBM-bound kernels and readbacks run underneath the other's VALU-bound marches.  n_workers 1..4 (default 2).
 * A further worker is only used while every worker still gets full-size batches and the call owns >= min_paths camera
 * paths (default 2^22); min_paths = 0 forces n_workers (tests). */
int rayn_hip_set_workers(rayn_ctx* ctx, int n_workers, uint64_t min_paths);
/* upper limit of the path-pool capacity per worker and batch of tiles (default 2^28 paths, ~89 GB of HBM per worker for a
 * scene without volume); the effective size is also capped so that all workers together use <= 60 % of the free HBM and that
 * the 32-bit [light sample][slot] references of a batch do not overflow. */
int rayn_hip_set_batch_paths(rayn_ctx* ctx, uint64_t paths);
/* First frame of a context (no reference counterpart; the reference renders ONE frame per process, src/main.rs:47-96).  Device
 * memory that a process released is wiped by the driver before another process can use it; a fresh process that takes a
 * full-size arena (up to 60 % of the HBM) right after another one exited runs its first frame up to 3.5 s longer (measured,
 * profiles/r03_cold_start.txt).  The FIRST frame a context renders therefore sizes its batches for arenas of at most `bytes`
 * in total (default 44 GB: one worker with 2^26-path batches of the volume path, 2^27 without - a quarter of the full size,
 * 2.5 % slower); from its second frame on a context uses full-size batches.  The
 * result is bit-identical (batching never changes a pixel: packets are per tile).  0 = full size from the first frame.
 *
 * Environment.  The library reads NO tuning from the environment unless RAYN_HIP_ENV_TUNING=1 is exported at context creation
 * (the measurement scripts under tools/ do); then RAYN_HIP_WORKERS, RAYN_HIP_WORKER_MIN_PATHS, RAYN_HIP_BATCH_PATHS,
 * RAYN_HIP_COLD_BYTES, RAYN_HIP_PROFILE (per-launch HIP events = rayn_hip_set_profiling), RAYN_HIP_REFILL_EXTEND / _SHADOW,
 * RAYN_HIP_PREFETCH_EXTEND / _SHADOW, RAYN_HIP_FAST_PATH and RAYN_HIP_PERSISTENT_BLOCKS preset the corresponding launch
 * parameters of a new context.  None of them changes a pixel. */
int rayn_hip_set_cold_bytes(rayn_ctx* ctx, uint64_t bytes);
/* mul_add policy (include/rayn_detmath.h): 0 = unfused a*b+c, what rayn's default x86-64 build does (wide
 * 0.4.6 without +fma) — the DEFAULT; 1 = fused, what rayn built with -C target-feature=+fma does.
 * rayn_hip_fma_policy() returns the default policy of a new ctx. */
int rayn_hip_fma_policy(void);
int rayn_hip_set_fma_policy(rayn_ctx* ctx, int policy);
/* sizeof() of the ABI structs as compiled: 0 world_desc, 1 frame_params, 2 stats, 3 hitable,
 * 4 material, 5 light, 6 camera — lets a binding verify its layout. */
size_t rayn_hip_sizeof(int which);
/* "" for the product build of the library; the VARIANT name of a `make variant` build (timing experiments: such a build is
 * only ever loaded through RAYN_HIP_LIB + RAYN_HIP_ALLOW_VARIANT=1, and bench.py prints the name in its result line). */
const char* rayn_hip_build_variant(void);

/* Restrict the following renders to the listed tiles (indices in the reference's tile order, src/film.rs:399-427; duplicates
 * and out-of-range indices are an error); n = 0 clears the restriction.  While a subset is set tile_first/tile_step are
 * ignored.  Tiles are independent (src/film.rs:439-627), so a subset render produces exactly the pixels the full frame
 * would: the parity tests use it to compare whole 16x16 tiles at the full BASELINE configurations with oracle digests. */
int rayn_hip_set_tile_subset(rayn_ctx* ctx, const uint32_t* tiles, uint32_t n);

/* Packet-order dump of ONE tile of the next renders (diagnostics; tile_index in the reference's tile order, -1 = off):
 * after every depth's bin stage the tile's binned queue is read back as records of 6 u32 {depth, object, tile x,
 * tile y, sample, valid} in HitStore::process_hits order (src/hitable.rs:94-134: object-major, insertion order, bins
 * padded to x4 with invalid lanes).  rayn_hip_get_trace returns the record count and copies up to cap_records. */
int rayn_hip_set_trace_tile(rayn_ctx* ctx, int tile_index);
int64_t rayn_hip_get_trace(const rayn_ctx* ctx, uint32_t* out, uint64_t cap_records);

/* ---- test probes on caller data (HOST pointers), so that the parity tests can compare single functions with the CPU oracle lane for lane:
 *   rayn_hip_probe_sdf_dist   SDF::dist (src/sdf.rs:125-140) of one TracedSDF, a per-lane device function;
 *   rayn_hip_probe_extend     HitableStore::add_hits' closest hit (src/hitable.rs:177-198 over Sphere::hit and TracedSDF::hit, src/sdf.rs:59-83) through the
 *                             PRODUCT extend kernel of the uploaded scene (k_extend1, or the generic k_extend of a multi-SDF scene) on a synthetic ray queue of
 *                             the n rays, ray time 0; out_obj 0xFFFFFFFF = none;
 *   rayn_hip_probe_shadow     the TracedSDF::occluded factors of HitableStore::test_occluded (src/hitable.rs:164-168, src/sdf.rs:25-57; the analytic spheres are
 *                             resolved by the shading kernel) through the PRODUCT shadow-march kernel (k_shadow1, k_shadow_bulb, the generic k_shadow) on a
 *                             synthetic job list of the n segments: 1.0 visible, 0.0 occluded;
 *   rayn_hip_probe_detmath    the pinned elementary functions (op 0 exp, 1 sin, 2 cos, 3 tan, 4 atan2(a,b), 5 pow(a,b)).
 *   Ops 6.. check the kernels' exact replacements of IEEE '/' and sqrt against the hardware IEEE
 *   result: 6 Newton-Raphson a/b, 7 IEEE a/b, 8 sqrt(a), 9/10/11 component x/y/z of v/|v| and
 *   12 |v| (a holds n xyz triples), 13 exhaustive sqrt sweep (out[i] = mismatch count over the
 *   65536 float bit patterns starting at bits(a[i])), 14 ln(a) as the Mandelbulb estimator evaluates it (dmf_logf),
 *   15 exhaustive sweep of 1 / sqrt(a) as the kernels evaluate it (rcp_sqrt_rn) against the IEEE sqrt and division (mismatch count over 65536 patterns), 16 that value. */
int rayn_hip_probe_sdf_dist(rayn_ctx* ctx, const rayn_frame_params* p, uint32_t hitable_index,
                            const float* pts_xyz, float* out, uint32_t n);
int rayn_hip_probe_extend(rayn_ctx* ctx, const rayn_frame_params* p, uint32_t depth,
                          const float* org_xyz, const float* dir_xyz, float* out_t,
                          uint32_t* out_obj, uint32_t n);
int rayn_hip_probe_shadow(rayn_ctx* ctx, const rayn_frame_params* p, const float* start_xyz,
                          const float* end_xyz, float* out, uint32_t n);
int rayn_hip_probe_detmath(rayn_ctx* ctx, uint32_t op, const float* a, const float* b, float* out,
                           uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RAYN_HIP_H */
