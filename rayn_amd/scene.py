"""Host-side mirror of rayn's scene trait surface, flattened to the POD `rayn_world_desc`.

Names, argument order and meaning follow the reference so that a scene written for rayn's
`setup.rs` reads the same here:
  Sphere::new(transform_seq, radius, material)            src/sphere.rs:14-20
  TracedSDF::new(sdf, material)                           src/sdf.rs:17-21
  MandelBox::new(iterations, box_fold, sphere_fold, scale) src/sdf.rs:114-122
  BoxFold::new(side_length) / SphereFold::new(min_radius, fixed_radius)  src/sdf.rs:151,172
  Dielectric::new_remap(albedo, roughness)                src/material.rs:167-174
  Lambertian::new / Sky::new / Emissive::new_splat        src/material.rs:97,401,464
  SphereLight::new(pos, rad, emission)                    src/light.rs:27-33
  PinholeCamera/ThinLensCamera/OrthographicCamera::new    src/camera.rs:53,134,228
  VolumeParams / World                                    src/volume.rs:1-5, src/world.rs:7-13
All arithmetic that the reference does in f32 at scene-construction time (colour scaling,
normalisation, roughness remap) is done here in numpy float32, in the same operation order.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi

f32 = np.float32


def vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def _mul(v, s):  # Vec3 * f32
    return (np.asarray(v, dtype=np.float32) * f32(s)).astype(np.float32)


def _fmaf(a, b, c):
    """IEEE fusedMultiplyAdd in binary32, exactly: the product of two binary32 values is exact in binary64 (48 significant bits),
    the sum with c is formed exactly as a ratio of integers and rounded ONCE to binary32 (no double rounding through binary64)."""
    from fractions import Fraction
    a, b, c = float(f32(a)), float(f32(b)), float(f32(c))
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        return f32(a * b + c)
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    # beyond the binary32 range IEEE fmaf returns +-inf (round to nearest: from max + half an ulp of max on), like the Rust scalar
    fmax = Fraction(float(np.finfo(np.float32).max))
    if abs(exact) >= fmax + Fraction(2) ** (127 - 24):
        return f32(np.inf) if exact > 0 else f32(-np.inf)
    if abs(exact) > fmax:
        return f32(np.finfo(np.float32).max) if exact > 0 else f32(-np.finfo(np.float32).max)
    lo = f32(float(exact))  # rounded through binary64: right except when it sits next to a binary32 rounding tie
    cands = sorted({lo, np.nextafter(lo, f32(-np.inf)), np.nextafter(lo, f32(np.inf))}, key=float)
    best = min(cands, key=lambda x: (abs(Fraction(float(x)) - exact), int(np.float32(x).view(np.uint32)) & 1))  # nearest, ties to even
    return f32(best)


def normalized(v):
    """ultraviolet's scalar Vec3::normalized (oracle assumptions A4 + A9): mag_sq = x.mul_add(x, y.mul_add(y, z*z)), then
    v * (1 / sqrt(mag_sq)).  Rust's scalar f32::mul_add is ALWAYS fused (libm fmaf without an FMA unit), whatever the build
    policy of the wide types is - src/setup.rs:100-101 normalises the light colours of the shipped scene this way (all of its
    products happen to be exact, so the constants are the same either way; an arbitrary scene's are not)."""
    v = np.asarray(v, dtype=np.float32)
    m2 = _fmaf(v[0], v[0], _fmaf(v[1], v[1], f32(v[2] * v[2])))
    r_mag = f32(1.0) / np.sqrt(f32(m2), dtype=np.float32)
    return (v * r_mag).astype(np.float32)


class Srgb:
    """src/spectrum.rs:5-75 — RGB newtype over Vec3."""

    def __init__(self, r, g, b):
        self.v = vec3(r, g, b)

    def __mul__(self, s):
        o = Srgb(0, 0, 0)
        o.v = _mul(self.v, s)
        return o

    def normalized(self):
        o = Srgb(0, 0, 0)
        o.v = normalized(self.v)
        return o


def _v(x):
    return x.v if isinstance(x, Srgb) else np.asarray(x, dtype=np.float32)


# ---- SDFs ---------------------------------------------------------------------------------
@dataclass
class BoxFold:
    side_length: float


@dataclass
class SphereFold:
    min_radius: float
    fixed_radius: float


@dataclass
class MandelBox:
    """MandelBox::new(iterations, box_fold, sphere_fold, scale), src/sdf.rs:114-122.  EXTENSION: scale_vel != 0 makes the scale the
    closure |t| scale + scale_vel * t (lane-0 time of the calling packet) - a fractal that morphs during the shutter."""
    iterations: int
    box_fold: BoxFold
    sphere_fold: SphereFold
    scale: float
    scale_vel: float = 0.0


@dataclass
class Mandelbulb:
    """EXTENSION (not in the reference): power-8 Mandelbulb DE, polynomial form, bailout 256."""
    iterations: int = 8


@dataclass
class SphereSDF:
    """sdfu::Sphere::new(radius) — the single-sphere SDF of BASELINE config 1."""
    radius: float


# ---- Hitables -----------------------------------------------------------------------------
@dataclass
class Sphere:
    transform_seq: np.ndarray  # constant Vec3 centre
    radius: float
    material: int


@dataclass
class TracedSDF:
    """TracedSDF::new(sdf, material), src/sdf.rs:17-21.  transform_seq is an EXTENSION with Sphere's semantics (a
    constant vec3 or Linear(base, vel): `|t| base + vel * t` at the packet's lane-0 time); None = the reference."""
    sdf: object
    material: int
    transform_seq: object = None


class HitableStore(list):
    def push(self, h):
        self.append(h)


# ---- Materials ----------------------------------------------------------------------------
@dataclass
class Lambertian:
    albedo: Srgb


@dataclass
class Dielectric:
    albedo: Srgb
    exponent: float  # already remapped

    @staticmethod
    def new_remap(albedo: Srgb, roughness: float):
        r = f32(1.0) - f32(roughness)
        e = f32(1.0) + f32(f32(f32(f32(r * r) * r) * r) * f32(300.0))
        return Dielectric(albedo, float(e))


@dataclass
class Sky:
    top: Srgb
    bottom: Srgb


@dataclass
class Emissive:
    emission: Srgb

    @staticmethod
    def new_splat(emission: Srgb):
        return Emissive(emission)


class MaterialStore(list):
    def add_material(self, m) -> int:  # -> MaterialHandle
        self.append(m)
        return len(self) - 1


@dataclass
class Linear:
    """The closure `move |t| base + vel * t` (a `Fn(f32) -> Vec3`, src/animation.rs:55-68) as data.  Usable for
    a camera's origin / at / up / focus; evaluated at lane 0's time of the ray-gen packet like the reference."""
    base: np.ndarray
    vel: np.ndarray


# ---- Lights / cameras / volume --------------------------------------------------------------
@dataclass
class SphereLight:
    pos: np.ndarray
    rad: float
    emission: Srgb


@dataclass
class PinholeCamera:
    resolution: Sequence[float]
    vfov: float
    origin: np.ndarray
    at: np.ndarray
    up: np.ndarray


@dataclass
class ThinLensCamera:
    resolution: Sequence[float]
    vfov: float
    aperture: float
    origin: np.ndarray
    at: np.ndarray
    up: np.ndarray
    focus: np.ndarray


@dataclass
class OrthographicCamera:
    resolution: Sequence[float]
    vertical_size: float
    origin: np.ndarray
    at: np.ndarray
    up: np.ndarray


class CameraStore(list):
    def add_camera(self, c) -> int:  # -> CameraHandle
        self.append(c)
        return len(self) - 1

    def get(self, handle: int):
        return self[handle]


@dataclass
class VolumeParams:
    coeff_scattering: Optional[float] = None
    coeff_extinction: Optional[float] = None


@dataclass
class World:
    hitables: HitableStore
    lights: List[SphereLight]
    materials: MaterialStore
    cameras: CameraStore
    volume_params: VolumeParams = field(default_factory=VolumeParams)

    def to_desc(self, camera: int) -> _abi.WorldDesc:
        """Flatten to the C-ABI descriptor (what a Rust shim would do for rayn's World)."""
        if len(self.hitables) > _abi.MAX_HITABLES or len(self.materials) > _abi.MAX_MATERIALS or len(self.lights) > _abi.MAX_LIGHTS:
            raise ValueError("scene exceeds the C-ABI fixed capacities")
        d = _abi.WorldDesc()
        d.n_hitables, d.n_materials, d.n_lights = len(self.hitables), len(self.materials), len(self.lights)

        def put(dst, v):
            v = _v(v)
            dst.x, dst.y, dst.z = float(v[0]), float(v[1]), float(v[2])

        for i, h in enumerate(self.hitables):
            o = d.hitables[i]
            o.material = h.material
            if isinstance(h, Sphere):
                o.kind = _abi.HITABLE_SPHERE
                if isinstance(h.transform_seq, Linear):
                    put(o.center, h.transform_seq.base)
                    put(o.center_vel, h.transform_seq.vel)
                    o.animated = 1
                else:
                    put(o.center, h.transform_seq)
                o.radius = h.radius
            elif isinstance(h, TracedSDF):
                o.kind = _abi.HITABLE_TRACED_SDF
                if isinstance(h.transform_seq, Linear):
                    put(o.center, h.transform_seq.base)
                    put(o.center_vel, h.transform_seq.vel)
                    o.animated = 1
                elif h.transform_seq is not None:
                    put(o.center, h.transform_seq)
                s = h.sdf
                if isinstance(s, MandelBox):
                    o.sdf_kind = _abi.SDF_MANDELBOX
                    o.iterations = s.iterations
                    o.box_side = s.box_fold.side_length
                    o.min_radius, o.fixed_radius = s.sphere_fold.min_radius, s.sphere_fold.fixed_radius
                    o.scale = s.scale
                    o.scale_vel = s.scale_vel
                elif isinstance(s, SphereSDF):
                    o.sdf_kind = _abi.SDF_SPHERE
                    o.sdf_radius = s.radius
                elif isinstance(s, Mandelbulb):
                    o.sdf_kind = _abi.SDF_MANDELBULB
                    o.iterations = s.iterations
                else:
                    raise TypeError(f"SDF {type(s).__name__} is outside the closed set")
            else:
                raise TypeError(f"hitable {type(h).__name__} is outside the closed set")
        for i, m in enumerate(self.materials):
            o = d.materials[i]
            if isinstance(m, Lambertian):
                o.kind = _abi.MAT_LAMBERTIAN
                put(o.a, m.albedo)
            elif isinstance(m, Dielectric):
                o.kind = _abi.MAT_DIELECTRIC
                put(o.a, m.albedo)
                o.exponent = m.exponent
            elif isinstance(m, Sky):
                o.kind = _abi.MAT_SKY
                put(o.a, m.top)
                put(o.b, m.bottom)
            elif isinstance(m, Emissive):
                o.kind = _abi.MAT_EMISSIVE
                put(o.a, m.emission)
            else:
                raise TypeError(f"material {type(m).__name__} is outside the closed set")
        for i, l in enumerate(self.lights):
            o = d.lights[i]
            put(o.pos, l.pos)
            o.rad = l.rad
            put(o.emission, l.emission)
        cam = self.cameras.get(camera)
        c = d.camera
        c.res_w, c.res_h = float(cam.resolution[0]), float(cam.resolution[1])

        def put_seq(dst, vel_dst, bit, v):
            if isinstance(v, Linear):
                put(dst, v.base)
                put(vel_dst, v.vel)
                c.animated |= 1 << bit
            else:
                put(dst, v)

        put_seq(c.origin, c.origin_vel, 0, cam.origin)
        put_seq(c.at, c.at_vel, 1, cam.at)
        put_seq(c.up, c.up_vel, 2, cam.up)
        if isinstance(cam, PinholeCamera):
            c.kind, c.vfov_or_size = _abi.CAM_PINHOLE, cam.vfov
        elif isinstance(cam, ThinLensCamera):
            c.kind, c.vfov_or_size, c.aperture = _abi.CAM_THIN_LENS, cam.vfov, cam.aperture
            put_seq(c.focus, c.focus_vel, 3, cam.focus)
        elif isinstance(cam, OrthographicCamera):
            c.kind, c.vfov_or_size = _abi.CAM_ORTHOGRAPHIC, cam.vertical_size
        else:
            raise TypeError(f"camera {type(cam).__name__} is outside the closed set")
        vp = self.volume_params
        d.has_scattering = 0 if vp.coeff_scattering is None else 1
        d.coeff_scattering = vp.coeff_scattering or 0.0
        d.has_extinction = 0 if vp.coeff_extinction is None else 1
        d.coeff_extinction = vp.coeff_extinction or 0.0
        return d


# ---- integrator / filter parameter holders -----------------------------------------------------
@dataclass
class PathTracingIntegrator:
    """src/integrator.rs:32-45."""
    max_bounces: int
    volume_marches: int = 2

    def requested_1d_sample_sets(self):
        return (self.max_bounces + 1) * (3 + self.volume_marches)

    def requested_2d_sample_sets(self):
        return (self.max_bounces + 1) * (12 + 8 * self.volume_marches)


@dataclass
class BlackmanHarrisFilter:
    """src/filter.rs:12-49."""
    radius: float = 1.5
    kind: int = 0


@dataclass
class BoxFilter:
    """src/filter.rs:110-140."""
    radius: float = 0.5
    kind: int = 1


@dataclass
class MitchellNetravaliFilter:
    """src/filter.rs:51-108 (Default: radius 2, b = c = 1/3)."""
    radius: float = 2.0
    b: float = 1.0 / 3.0
    c: float = 1.0 / 3.0
    kind: int = 2

    @property
    def params(self):
        return (self.b, self.c)


@dataclass
class LanczosSincFilter:
    """src/filter.rs:142-185 (Default: radius 3, tau 3)."""
    radius: float = 3.0
    tau: float = 3.0
    kind: int = 3

    @property
    def params(self):
        return (self.tau, 0.0)
