"""Scene definitions: the reference's `setup::setup()` (src/setup.rs:46-170) and the BASELINE
config scenes derived from it (S0, S1, S2).  Constants mirror
src/setup.rs:16-44 and src/main.rs:47-56,69."""
import numpy as np

from .scene import (BoxFold, CameraStore, Dielectric, Emissive, HitableStore, MandelBox, Mandelbulb, MaterialStore,
                    PinholeCamera, Sky, Sphere, SphereFold, SphereLight, SphereSDF, Srgb, TracedSDF,
                    VolumeParams, World, vec3, _mul)

WORLD_RADIUS = 100.0          # src/setup.rs:33
SDF_DETAIL_SCALE = 0.5        # src/setup.rs:37
FRACTAL_ITERATIONS = 12       # src/setup.rs:44
VOLUME_MARCHES_PER_SAMPLE = 2  # src/setup.rs:25
MAX_MARCHES = 256             # src/sdf.rs:9
MAX_VIS_MARCHES = 100         # src/sdf.rs:10
FRAME_RATE = 24               # src/main.rs:47
TILE_SIZE = (16, 16)          # src/main.rs:69
FILTER_RADIUS = 1.5           # src/main.rs:51


def _lights_and_proxies(materials, hitables, lights, with_center_light):
    green = Srgb(1.5, 4.5, 3.0).normalized()
    blue = Srgb(1.5, 3.0, 4.5).normalized()
    blue_emissive = materials.add_material(Emissive.new_splat(blue * 3.0))
    green_emissive = materials.add_material(Emissive.new_splat(green * 3.0))
    light_pairs = [(vec3(1.2, -1.2, 1.2), 0.15), (vec3(-1.2, 1.2, 1.2), 0.15)]
    for pos, rad in light_pairs:
        green_pos = pos.copy()
        green_pos[1] *= -1.0
        lights.append(SphereLight(green_pos, rad, green * 40.0))
        lights.append(SphereLight(pos, rad, blue * 40.0))
        hitables.push(Sphere(green_pos, float(np.float32(rad) - np.float32(0.01)), green_emissive))
        hitables.push(Sphere(pos, float(np.float32(rad) - np.float32(0.01)), blue_emissive))
    if with_center_light:
        lights.append(SphereLight(vec3(0.0, 0.0, 0.0), 0.25, green * 20.0))
        hitables.push(Sphere(vec3(0.0, 0.0, 0.0), 0.24, green_emissive))


def _camera(resolution, distance=2.25):
    cameras = CameraStore()
    cam = PinholeCamera((float(resolution[0]), float(resolution[1])), 60.0,
                        _mul(vec3(-0.45, 0.2, 2.0), distance), vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0))
    return cameras, cameras.add_camera(cam)


def setup(resolution=(1280, 720), volumes=True, sdf="mandelbox"):
    """`setup::setup()` (src/setup.rs:46-170).  volumes=True is the shipped scene (S2: rho_s 0.25,
    rho_t 0.035); volumes=False is S1 (both None).  sdf="sphere" swaps the fractal for a unit
    sphere SDF and drops the central light (S0, BASELINE config 1)."""
    materials, hitables, lights = MaterialStore(), HitableStore(), []
    volume_params = VolumeParams(0.25, 0.035) if volumes else VolumeParams(None, None)
    sky = materials.add_material(Sky(Srgb(0.3, 0.4, 0.6), Srgb(0.2, 0.3, 0.6) * 0.05))
    hitables.push(Sphere(vec3(0.0, 0.0, 0.0), WORLD_RADIUS, sky))
    grey = materials.add_material(Dielectric.new_remap(Srgb(0.2, 0.2, 0.2), 0.6))
    if sdf == "mandelbox":
        hitables.push(TracedSDF(MandelBox(FRACTAL_ITERATIONS, BoxFold(1.0), SphereFold(0.01, 1.9), -2.1), grey))
    elif sdf == "sphere":
        hitables.push(TracedSDF(SphereSDF(1.0), grey))
    elif sdf == "mandelbulb":  # extension: the fractal BASELINE.json names; the reference has none
        hitables.push(TracedSDF(Mandelbulb(8), grey))
    else:
        raise ValueError(sdf)
    _lights_and_proxies(materials, hitables, lights, with_center_light=(sdf == "mandelbox"))
    cameras, camera = _camera(resolution, 1.35 if sdf == "mandelbulb" else 2.25)  # the bulb is ~1.2 units in radius
    return camera, World(hitables, lights, materials, cameras, volume_params)


def setup_s0(resolution=(256, 256)):
    """BASELINE config 1: single-sphere SDF scene, volumes off."""
    return setup(resolution, volumes=False, sdf="sphere")


def setup_s1(resolution=(1920, 1080)):
    """BASELINE configs 2/4: shipped MandelBox scene, volumes off."""
    return setup(resolution, volumes=False, sdf="mandelbox")


def setup_bulb(resolution=(1920, 1080), volumes=False):
    """EXTENSION scene: the shipped scene with the MandelBox swapped for a power-8 Mandelbulb (no central light,
    which would sit inside the bulb)."""
    return setup(resolution, volumes=volumes, sdf="mandelbulb")


def setup_bulb3(resolution=(1920, 1080)):
    """EXTENSION scene of the metric's literally named workload ("1920x1080 Mandelbulb @1024spp" on configs[2]): the Mandelbulb
    scene WITH the shipped homogeneous volume (rho_s 0.25, rho_t 0.035, src/setup.rs:58-59)."""
    return setup(resolution, volumes=True, sdf="mandelbulb")


def setup_s2(resolution=(1920, 1080)):
    """BASELINE config 3: shipped MandelBox scene with the homogeneous volume."""
    return setup(resolution, volumes=True, sdf="mandelbox")


def setup_s3(resolution=(7680, 4320), moving_fractal=True):
    """BASELINE config 5's scene: S1 seen by a camera whose origin is a closure of time
    `|t| origin + vel * t` (src/animation.rs:55-68: every packet evaluates it at lane 0's time) -> time-sampled motion blur
    over the shutter interval.  The reference's TracedSDF itself ignores time (src/sdf.rs:25); with moving_fractal the
    fractal also translates (our TracedSDF.transform_seq extension) - "animated fractal with time-sampled motion blur"."""
    from .scene import Linear
    camera, world = setup(resolution, volumes=False, sdf="mandelbox")
    cam = world.cameras.get(camera)
    cam.origin = Linear(cam.origin, vec3(0.9, -0.3, 0.15))
    if moving_fractal:  # EXTENSION (rayn_hip.h, rayn_hitable.animated): the fractal itself translates during the shutter
        for h in world.hitables:
            if isinstance(h, TracedSDF):
                h.transform_seq = Linear(vec3(0.0, 0.0, 0.0), vec3(-0.6, 0.45, 0.3))
    return camera, world


def setup_bulb5(resolution=(7680, 4320)):
    """EXTENSION scene of BASELINE configs[4] as literally named ("animated Mandelbulb with time-sampled motion blur"): the
    Mandelbulb scene (volumes off) seen by S3's moving camera (reference-supported closure, src/animation.rs:55-68) while the bulb
    itself translates during the shutter (TracedSDF.transform_seq extension -> rayn_hitable.center_vel).  Nothing in rayn
    corresponds: its only fractal is the MandelBox (src/sdf.rs:104-141) and its TracedSDF ignores time (src/sdf.rs:25,59)."""
    from .scene import Linear
    camera, world = setup(resolution, volumes=False, sdf="mandelbulb")
    cam = world.cameras.get(camera)
    cam.origin = Linear(cam.origin, vec3(0.54, -0.18, 0.09))  # S3's camera drift scaled by the bulb scene's shorter camera distance (1.35 / 2.25)
    for h in world.hitables:
        if isinstance(h, TracedSDF):
            h.transform_seq = Linear(vec3(0.0, 0.0, 0.0), vec3(-0.36, 0.27, 0.18))
    return camera, world


# scene tag -> constructor (bench.py WORKLOADS, tests/golden/make_config_digests.py CONFIGS, tools/): "ship" = setup::setup() as shipped
SCENES = {"s0": setup_s0, "s1": setup_s1, "s2": setup_s2, "s3": setup_s3, "bulb": setup_bulb, "bulbv": setup_bulb3, "bulbm": setup_bulb5, "ship": setup}
