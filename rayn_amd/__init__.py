"""rayn_amd — MI355X-native (gfx950) wavefront implementation of fu5ha/rayn's per-sample
integrator hot path, behind rayn's own Film / World / Hitable / Material / Light / Camera surface.

The compute path is rayn_amd/csrc/librayn_hip.so (hand-written HIP, C ABI in include/rayn_hip.h);
this package is the thin host mirror.  There is no CPU fallback."""
from .film import ChannelKind, Context, Film, build_tables  # noqa: F401
from .params import frame_params  # noqa: F401
from .scene import (BlackmanHarrisFilter, BoxFilter, LanczosSincFilter, MitchellNetravaliFilter, BoxFold, CameraStore, Dielectric, Emissive, HitableStore, Lambertian, Linear,  # noqa: F401
                    MandelBox, Mandelbulb, MaterialStore, OrthographicCamera, PathTracingIntegrator, PinholeCamera, Sky, Sphere, SphereFold,
                    SphereLight, SphereSDF, Srgb, ThinLensCamera, TracedSDF, VolumeParams, World, vec3)
from . import setup  # noqa: F401
