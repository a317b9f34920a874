// rayn_host.hpp — C++ host-side mirror of rayn's scene/film API on top of the C ABI (rayn_hip.h).
//
// The reference is compiled Rust; its toolchain is absent here, so this header is the compiled-language
// host side: the same type and method names, argument order and meaning as the reference, flattening to
// the POD `rayn_world_desc` exactly where a Rust shim would.  Header-only; link with
// -lrayn_hip.  All construction-time arithmetic is f32 in the reference's operation order (compile with
// -ffp-contract=off if bit-equality of the scene constants with another host matters).
//
//   reference                                              here
//   Srgb::new / * f32 / normalized   src/spectrum.rs        rayn::Srgb
//   Sphere::new(center, r, mat)      src/sphere.rs:14-20    rayn::Sphere
//   TracedSDF::new(sdf, mat)         src/sdf.rs:17-21       rayn::TracedSDF
//   MandelBox::new / BoxFold / SphereFold  src/sdf.rs:114,151,172
//   Lambertian / Dielectric::new_remap / Sky::new / Emissive::new_splat   src/material.rs
//   MaterialStore::add_material -> MaterialHandle            src/material.rs:55-73
//   SphereLight::new                 src/light.rs:27-33
//   Pinhole/ThinLens/OrthographicCamera::new, CameraStore   src/camera.rs
//   VolumeParams, World              src/volume.rs, src/world.rs
//   PathTracingIntegrator            src/integrator.rs:32-45
//   BlackmanHarrisFilter / BoxFilter / MitchellNetravaliFilter / LanczosSincFilter src/filter.rs:12-185
//   Film::new / render_frame_into    src/film.rs:184-203, 382-395
//   setup::setup()                   src/setup.rs:46-170
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "rayn_hip.h"

namespace rayn {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 zero() { return Vec3(0, 0, 0); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    // ultraviolet's SCALAR Vec3 (oracle assumption A4/A9): mag_sq = x.mul_add(x, y.mul_add(y, z * z)), v * (1 / mag).  Rust's scalar
    // f32::mul_add is ALWAYS a fused multiply-add (libm fmaf when the target has no FMA unit) - under both build policies of the
    // wide types - so the host mirror uses std::fmaf unconditionally (src/setup.rs:100-101 is the call site of the shipped scene).
    Vec3 normalized() const {
        float m2 = std::fmaf(x, x, std::fmaf(y, y, z * z));
        float r = 1.0f / std::sqrt(m2);
        return Vec3(x * r, y * r, z * r);
    }
    rayn_vec3 pod() const { return rayn_vec3{x, y, z}; }
};
struct Vec2 { float x = 0, y = 0; Vec2() = default; Vec2(float x_, float y_) : x(x_), y(y_) {} };
struct Extent2u { uint32_t w = 0, h = 0; Extent2u() = default; Extent2u(uint32_t w_, uint32_t h_) : w(w_), h(h_) {} };

struct Srgb { // src/spectrum.rs:5-75
    Vec3 v;
    Srgb() = default;
    Srgb(float r, float g, float b) : v(r, g, b) {}
    static Srgb new_(float r, float g, float b) { return Srgb(r, g, b); }
    Srgb operator*(float s) const { Srgb o; o.v = v * s; return o; }
    Srgb normalized() const { Srgb o; o.v = v.normalized(); return o; }
};

struct MaterialHandle { size_t idx = 0; };
struct CameraHandle { size_t idx = 0; };

// ---- materials ---------------------------------------------------------------------------------
struct Lambertian { Srgb albedo; static Lambertian new_(Srgb a) { return Lambertian{a}; } };
struct Dielectric {
    Srgb albedo; float exponent;
    // Roughness between 0.0 (smooth) and 1.0 (rough); src/material.rs:167-174
    static Dielectric new_remap(Srgb albedo, float roughness) {
        float r = 1.0f - roughness;
        float e = 1.0f + r * r * r * r * 300.0f;
        return Dielectric{albedo, e};
    }
};
struct Sky { Srgb top, bottom; static Sky new_(Srgb t, Srgb b) { return Sky{t, b}; } };
struct Emissive { Srgb emission; static Emissive new_splat(Srgb e) { return Emissive{e}; } };
using Material = std::variant<Lambertian, Dielectric, Sky, Emissive>;

class MaterialStore {
  public:
    MaterialHandle add_material(Material m) { mats.push_back(std::move(m)); return MaterialHandle{mats.size() - 1}; }
    size_t len() const { return mats.size(); }
    std::vector<Material> mats;
};

// ---- SDFs + hitables ----------------------------------------------------------------------------
struct BoxFold { float side_length; static BoxFold new_(float s) { return BoxFold{s}; } };
struct SphereFold { float min_radius, fixed_radius; static SphereFold new_(float a, float b) { return SphereFold{a, b}; } };
struct MandelBox {
    uint32_t iterations; BoxFold box_fold; SphereFold sphere_fold; float scale;
    float scale_vel = 0.0f; // EXTENSION: scale(t) = scale + scale_vel * t at lane-0 time (rayn_hip.h); 0 = the reference
    static MandelBox new_(uint32_t it, BoxFold b, SphereFold s, float scale) { MandelBox m{it, b, s, scale}; return m; }
};
struct SphereSDF { float radius; }; // sdfu::Sphere::new(radius)
using SDF = std::variant<MandelBox, SphereSDF>;

// A time-sequenced Vec3 (src/animation.rs): a constant, or the closure `move |t| base + vel * t`
struct Sequenced3 {
    Vec3 base, vel; bool animated = false;
    Sequenced3() = default;
    Sequenced3(Vec3 constant) : base(constant) {}
    static Sequenced3 linear(Vec3 base, Vec3 vel) { Sequenced3 s; s.base = base; s.vel = vel; s.animated = true; return s; }
};

struct Sphere { Sequenced3 transform_seq; float radius; MaterialHandle material;
                static Sphere new_(Sequenced3 c, float r, MaterialHandle m) { return Sphere{c, r, m}; } };
// transform_seq is an EXTENSION (the reference's TracedSDF has none, src/sdf.rs:12-21): Sphere's transform semantics for
// an SDF - constant or |t| base + vel * t sampled at the packet's lane-0 time; default = the reference's behaviour.
struct TracedSDF { SDF sdf; MaterialHandle material; Sequenced3 transform_seq;
                   static TracedSDF new_(SDF s, MaterialHandle m) { return TracedSDF{std::move(s), m, Sequenced3()}; }
                   static TracedSDF new_moving(SDF s, MaterialHandle m, Sequenced3 tr) { return TracedSDF{std::move(s), m, tr}; } };
using Hitable = std::variant<Sphere, TracedSDF>;

class HitableStore {
  public:
    void push(Hitable h) { items.push_back(std::move(h)); }
    size_t len() const { return items.size(); }
    std::vector<Hitable> items;
};

// ---- lights, cameras, volume --------------------------------------------------------------------
struct SphereLight { Vec3 pos; float rad; Srgb emission; static SphereLight new_(Vec3 p, float r, Srgb e) { return SphereLight{p, r, e}; } };

struct PinholeCamera { Vec2 resolution; float vfov; Sequenced3 origin, at, up;
                       static PinholeCamera new_(Vec2 res, float vfov, Sequenced3 o, Sequenced3 a, Sequenced3 u) { return PinholeCamera{res, vfov, o, a, u}; } };
struct ThinLensCamera { Vec2 resolution; float vfov, aperture; Sequenced3 origin, at, up, focus;
                        static ThinLensCamera new_(Vec2 res, float vfov, float ap, Sequenced3 o, Sequenced3 a, Sequenced3 u, Sequenced3 f) { return ThinLensCamera{res, vfov, ap, o, a, u, f}; } };
struct OrthographicCamera { Vec2 resolution; float vertical_size; Sequenced3 origin, at, up;
                            static OrthographicCamera new_(Vec2 res, float vs, Sequenced3 o, Sequenced3 a, Sequenced3 u) { return OrthographicCamera{res, vs, o, a, u}; } };
using Camera = std::variant<PinholeCamera, ThinLensCamera, OrthographicCamera>;
class CameraStore {
  public:
    CameraHandle add_camera(Camera c) { cams.push_back(std::move(c)); return CameraHandle{cams.size() - 1}; }
    const Camera& get(CameraHandle h) const { return cams.at(h.idx); }
    std::vector<Camera> cams;
};

struct VolumeParams { std::optional<float> coeff_scattering, coeff_extinction; };

struct World { // src/world.rs:7-13
    HitableStore hitables;
    std::vector<SphereLight> lights;
    MaterialStore materials;
    CameraStore cameras;
    VolumeParams volume_params;

    // flatten to the C-ABI descriptor; scene order is preserved (it is semantic, src/hitable.rs:170-210)
    rayn_world_desc to_desc(CameraHandle camera) const {
        if (hitables.len() > RAYN_MAX_HITABLES || materials.len() > RAYN_MAX_MATERIALS || lights.size() > RAYN_MAX_LIGHTS)
            throw std::invalid_argument("scene exceeds the C-ABI fixed capacities");
        rayn_world_desc d;
        std::memset(&d, 0, sizeof d);
        d.n_hitables = (uint32_t)hitables.len(); d.n_materials = (uint32_t)materials.len(); d.n_lights = (uint32_t)lights.size();
        for (size_t i = 0; i < hitables.len(); i++) {
            rayn_hitable& o = d.hitables[i];
            if (const Sphere* s = std::get_if<Sphere>(&hitables.items[i])) {
                o.kind = RAYN_HITABLE_SPHERE; o.material = (uint32_t)s->material.idx; o.center = s->transform_seq.base.pod(); o.radius = s->radius;
                if (s->transform_seq.animated) { o.animated = 1; o.center_vel = s->transform_seq.vel.pod(); }
            } else {
                const TracedSDF& t = std::get<TracedSDF>(hitables.items[i]);
                o.kind = RAYN_HITABLE_TRACED_SDF; o.material = (uint32_t)t.material.idx; o.center = t.transform_seq.base.pod();
                if (t.transform_seq.animated) { o.animated = 1; o.center_vel = t.transform_seq.vel.pod(); }
                if (const MandelBox* m = std::get_if<MandelBox>(&t.sdf)) {
                    o.sdf_kind = RAYN_SDF_MANDELBOX; o.iterations = m->iterations; o.box_side = m->box_fold.side_length;
                    o.min_radius = m->sphere_fold.min_radius; o.fixed_radius = m->sphere_fold.fixed_radius; o.scale = m->scale; o.scale_vel = m->scale_vel;
                } else { o.sdf_kind = RAYN_SDF_SPHERE; o.sdf_radius = std::get<SphereSDF>(t.sdf).radius; }
            }
        }
        for (size_t i = 0; i < materials.len(); i++) {
            rayn_material& o = d.materials[i];
            const Material& m = materials.mats[i];
            if (const Lambertian* l = std::get_if<Lambertian>(&m)) { o.kind = RAYN_MAT_LAMBERTIAN; o.a = l->albedo.v.pod(); }
            else if (const Dielectric* di = std::get_if<Dielectric>(&m)) { o.kind = RAYN_MAT_DIELECTRIC; o.a = di->albedo.v.pod(); o.exponent = di->exponent; }
            else if (const Sky* s = std::get_if<Sky>(&m)) { o.kind = RAYN_MAT_SKY; o.a = s->top.v.pod(); o.b = s->bottom.v.pod(); }
            else { o.kind = RAYN_MAT_EMISSIVE; o.a = std::get<Emissive>(m).emission.v.pod(); }
        }
        for (size_t i = 0; i < lights.size(); i++) {
            d.lights[i].pos = lights[i].pos.pod(); d.lights[i].rad = lights[i].rad; d.lights[i].emission = lights[i].emission.v.pod();
        }
        rayn_camera& c = d.camera;
        const Camera& cam = cameras.get(camera);
        auto put = [&c](rayn_vec3& dst, rayn_vec3& vel, uint32_t bit, const Sequenced3& s) { dst = s.base.pod(); if (s.animated) { vel = s.vel.pod(); c.animated |= 1u << bit; } };
        if (const PinholeCamera* p = std::get_if<PinholeCamera>(&cam)) {
            c.kind = RAYN_CAM_PINHOLE; c.res_w = p->resolution.x; c.res_h = p->resolution.y; c.vfov_or_size = p->vfov;
            put(c.origin, c.origin_vel, 0, p->origin); put(c.at, c.at_vel, 1, p->at); put(c.up, c.up_vel, 2, p->up);
        } else if (const ThinLensCamera* t = std::get_if<ThinLensCamera>(&cam)) {
            c.kind = RAYN_CAM_THIN_LENS; c.res_w = t->resolution.x; c.res_h = t->resolution.y; c.vfov_or_size = t->vfov; c.aperture = t->aperture;
            put(c.origin, c.origin_vel, 0, t->origin); put(c.at, c.at_vel, 1, t->at); put(c.up, c.up_vel, 2, t->up); put(c.focus, c.focus_vel, 3, t->focus);
        } else {
            const OrthographicCamera& o = std::get<OrthographicCamera>(cam);
            c.kind = RAYN_CAM_ORTHOGRAPHIC; c.res_w = o.resolution.x; c.res_h = o.resolution.y; c.vfov_or_size = o.vertical_size;
            put(c.origin, c.origin_vel, 0, o.origin); put(c.at, c.at_vel, 1, o.at); put(c.up, c.up_vel, 2, o.up);
        }
        if (volume_params.coeff_scattering) { d.has_scattering = 1; d.coeff_scattering = *volume_params.coeff_scattering; }
        if (volume_params.coeff_extinction) { d.has_extinction = 1; d.coeff_extinction = *volume_params.coeff_extinction; }
        return d;
    }
};

// ---- integrator + filters -----------------------------------------------------------------------
struct PathTracingIntegrator { // src/integrator.rs:32-45
    size_t max_bounces = 3, volume_marches = 2;
    size_t requested_1d_sample_sets() const { return (max_bounces + 1) * (3 + volume_marches); }
    size_t requested_2d_sample_sets() const { return (max_bounces + 1) * (12 + 8 * volume_marches); }
};
struct BlackmanHarrisFilter { float radius_ = 1.5f; static BlackmanHarrisFilter new_(float r) { return BlackmanHarrisFilter{r}; }
                              float radius() const { return radius_; } static constexpr uint32_t kind = 0; };
struct BoxFilter { float radius_ = 0.5f; static BoxFilter new_(float r) { return BoxFilter{r}; }
                   float radius() const { return radius_; } static constexpr uint32_t kind = 1; };
struct MitchellNetravaliFilter { // src/filter.rs:51-108
    float radius_ = 2.0f, b = 1.0f / 3.0f, c = 1.0f / 3.0f;
    static MitchellNetravaliFilter new_(float r, float b, float c) { return MitchellNetravaliFilter{r, b, c}; }
    float radius() const { return radius_; } float param0() const { return b; } float param1() const { return c; }
    static constexpr uint32_t kind = 2;
};
struct LanczosSincFilter { // src/filter.rs:142-185
    float radius_ = 3.0f, tau = 3.0f;
    static LanczosSincFilter new_(float r, float tau) { return LanczosSincFilter{r, tau}; }
    float radius() const { return radius_; } float param0() const { return tau; } float param1() const { return 0.0f; }
    static constexpr uint32_t kind = 3;
};
namespace detail { // parameter-free filters have no param0()/param1()
template <class F> auto filter_p0(const F& f, int) -> decltype(f.param0()) { return f.param0(); }
template <class F> float filter_p0(const F&, long) { return 0.0f; }
template <class F> auto filter_p1(const F& f, int) -> decltype(f.param1()) { return f.param1(); }
template <class F> float filter_p1(const F&, long) { return 0.0f; }
}

// constants the tile closure reads (src/setup.rs:16-44, src/sdf.rs:9-10)
struct RenderConstants { uint32_t max_marches = 256, max_vis_marches = 100; float sdf_detail_scale = 0.5f, world_radius = 100.0f; };

// ---- Film ----------------------------------------------------------------------------------------
enum class ChannelKind { Color, Alpha, Background, WorldNormal }; // src/film.rs:103-120

class Film {
  public:
    // Film::new (src/film.rs:184-203): duplicate channel kinds are an error
    Film(const std::vector<ChannelKind>& channels, Extent2u res, int device = 0) : kinds_(channels), res_(res) {
        for (size_t i = 0; i < channels.size(); i++)
            for (size_t j = i + 1; j < channels.size(); j++)
                if (channels[i] == channels[j]) throw std::invalid_argument("Attempted to create multiple channels of one kind");
        int rc = rayn_hip_create(device, &ctx_);
        if (rc != RAYN_OK) throw std::runtime_error("rayn_hip_create failed: " + std::to_string(rc) + " (no GPU? there is no CPU fallback)");
        const size_t n = (size_t)res.w * res.h;
        color.assign(3 * n, 0.0f); alpha.assign(n, 0.0f); background.assign(3 * n, 0.0f); world_normal.assign(3 * n, 0.0f);
    }
    // the same film rendered by several GPUs of this process (rayn_hip_create_multi: tiles dealt in rotation, one peer copy per GPU)
    Film(const std::vector<ChannelKind>& channels, Extent2u res, const std::vector<int>& devices) : kinds_(channels), res_(res) {
        for (size_t i = 0; i < channels.size(); i++)
            for (size_t j = i + 1; j < channels.size(); j++)
                if (channels[i] == channels[j]) throw std::invalid_argument("Attempted to create multiple channels of one kind");
        int rc = rayn_hip_create_multi(devices.data(), (int)devices.size(), &ctx_);
        if (rc != RAYN_OK) throw std::runtime_error("rayn_hip_create_multi failed: " + std::to_string(rc) + " (no GPU? there is no CPU fallback)");
        const size_t n = (size_t)res.w * res.h;
        color.assign(3 * n, 0.0f); alpha.assign(n, 0.0f); background.assign(3 * n, 0.0f); world_normal.assign(3 * n, 0.0f);
    }
    ~Film() { if (ctx_) rayn_hip_destroy(ctx_); }
    Film(const Film&) = delete;
    Film& operator=(const Film&) = delete;

    // Film::render_frame_into (src/film.rs:382-395); the film is overwritten, not accumulated (src/film.rs:91)
    template <typename F>
    void render_frame_into(const World& world, CameraHandle camera, const PathTracingIntegrator& integrator, const F& filter,
                           Extent2u tile_size, size_t frame, std::pair<float, float> time_range, size_t samples,
                           const RenderConstants& k = RenderConstants()) {
        rayn_world_desc desc = world.to_desc(camera);
        check(rayn_hip_upload_world(ctx_, &desc));
        rayn_frame_params p;
        std::memset(&p, 0, sizeof p);
        p.width = res_.w; p.height = res_.h; p.samples = (uint32_t)samples; p.tile_w = tile_size.w; p.tile_h = tile_size.h;
        p.max_bounces = (uint32_t)integrator.max_bounces; p.volume_marches = (uint32_t)integrator.volume_marches; p.frame = (uint32_t)frame;
        p.time_start = time_range.first; p.time_end = time_range.second;
        p.max_marches = k.max_marches; p.max_vis_marches = k.max_vis_marches; p.sdf_detail_scale = k.sdf_detail_scale; p.world_radius = k.world_radius;
        p.tile_first = 0; p.tile_step = 1;
        const uint32_t spp = 4u * (uint32_t)samples;
        const uint32_t sets_1d = 1 + (uint32_t)integrator.requested_1d_sample_sets(), sets_2d = 2 + (uint32_t)integrator.requested_2d_sample_sets(); // src/film.rs:431-432
        std::vector<float> s1((size_t)spp * sets_1d), s2((size_t)spp * 2 * sets_2d), scr((size_t)res_.w * res_.h), fis(RAYN_FIS_TABLE_SIZE);
        check(rayn_build_rd_tables(spp, sets_1d, sets_2d, frame, s1.data(), s2.data()));   // Samples::new_rd, src/film.rs:434
        check(rayn_build_scramble(res_.w, res_.h, scr.data()));                              // src/film.rs:460-461
        check(rayn_build_fis_table_ex(F::kind, filter.radius(), detail::filter_p0(filter, 0), detail::filter_p1(filter, 0), fis.data()));                   // src/film.rs:429
        check(rayn_hip_render_frame(ctx_, &p, s1.data(), s2.data(), scr.data(), fis.data(), color.data(), alpha.data(), background.data(),
                                    world_normal.data()));
        progressive_epoch++;
    }
    rayn_stats stats() const { rayn_stats s; rayn_hip_get_stats(ctx_, &s); return s; }
    // what GPU `entry` of a multi-device film did in the last frame (its tiles / segments / its own HIP-event time): rayn_hip_get_entry_stats
    int device_count() const { return rayn_hip_device_count(ctx_); }
    rayn_stats device_stats(int entry) const {
        rayn_stats s;
        if (rayn_hip_get_entry_stats(ctx_, entry, &s) != RAYN_OK) throw std::runtime_error("rayn_hip_get_entry_stats: no such entry");
        return s;
    }
    Extent2u res() const { return res_; }
    rayn_ctx* ctx() { return ctx_; }

    // channel storage, bottom-up rows like the reference (the flip happens in save_to, src/film.rs:236)
    std::vector<float> color, alpha, background, world_normal;
    size_t progressive_epoch = 0;

  private:
    void check(int rc) { if (rc != RAYN_OK) throw std::runtime_error(std::string("rayn_hip: ") + rayn_hip_last_error(ctx_)); }
    std::vector<ChannelKind> kinds_;
    Extent2u res_;
    rayn_ctx* ctx_ = nullptr;
};

// ---- setup::setup() (src/setup.rs:46-170) with the resolution as an argument ---------------------
namespace setup {
constexpr float WORLD_RADIUS = 100.0f;
constexpr uint32_t FRACTAL_ITERATIONS = 12;
inline std::pair<CameraHandle, World> setup(Extent2u resolution = Extent2u(1280, 720), bool volumes = true) {
    World w;
    if (volumes) w.volume_params = VolumeParams{0.25f, 0.035f};
    MaterialHandle sky = w.materials.add_material(Sky::new_(Srgb::new_(0.3f, 0.4f, 0.6f), Srgb::new_(0.2f, 0.3f, 0.6f) * 0.05f));
    w.hitables.push(Sphere::new_(Vec3(0.0f, 0.0f, 0.0f), WORLD_RADIUS, sky));
    MaterialHandle grey = w.materials.add_material(Dielectric::new_remap(Srgb::new_(0.2f, 0.2f, 0.2f), 0.6f));
    w.hitables.push(TracedSDF::new_(MandelBox::new_(FRACTAL_ITERATIONS, BoxFold::new_(1.0f), SphereFold::new_(0.01f, 1.9f), -2.1f), grey));
    Srgb green = Srgb::new_(1.5f, 4.5f, 3.0f).normalized();
    Srgb blue = Srgb::new_(1.5f, 3.0f, 4.5f).normalized();
    MaterialHandle blue_emissive = w.materials.add_material(Emissive::new_splat(blue * 3.0f));
    MaterialHandle green_emissive = w.materials.add_material(Emissive::new_splat(green * 3.0f));
    const std::pair<Vec3, float> light_pairs[2] = {{Vec3(1.2f, -1.2f, 1.2f), 0.15f}, {Vec3(-1.2f, 1.2f, 1.2f), 0.15f}};
    for (const auto& lp : light_pairs) {
        Vec3 pos = lp.first, green_pos = lp.first;
        float rad = lp.second;
        green_pos.y *= -1.0f;
        w.lights.push_back(SphereLight::new_(green_pos, rad, green * 40.0f));
        w.lights.push_back(SphereLight::new_(pos, rad, blue * 40.0f));
        w.hitables.push(Sphere::new_(green_pos, rad - 0.01f, green_emissive));
        w.hitables.push(Sphere::new_(pos, rad - 0.01f, blue_emissive));
    }
    w.lights.push_back(SphereLight::new_(Vec3::zero(), 0.25f, green * 20.0f));
    w.hitables.push(Sphere::new_(Vec3::zero(), 0.24f, green_emissive));
    Vec2 res((float)resolution.w, (float)resolution.h);
    CameraHandle cam = w.cameras.add_camera(PinholeCamera::new_(res, 60.0f, Vec3(-0.45f, 0.2f, 2.0f) * 2.25f, Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f)));
    return {cam, std::move(w)};
}
} // namespace setup

} // namespace rayn
