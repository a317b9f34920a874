"""ctypes mirror of include/rayn_hip.h (the C ABI).  Layouts must match the header byte for byte;
tests/test_abi.py checks the sizes against the compiled library."""
import ctypes as C

MAX_HITABLES = 16
MAX_MATERIALS = 16
MAX_LIGHTS = 16
FIS_TABLE_SIZE = 512

HITABLE_SPHERE, HITABLE_TRACED_SDF = 0, 1
SDF_SPHERE, SDF_MANDELBOX, SDF_MANDELBULB = 0, 1, 2
MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_SKY, MAT_EMISSIVE = 0, 1, 2, 3
CAM_PINHOLE, CAM_THIN_LENS, CAM_ORTHOGRAPHIC = 0, 1, 2


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Hitable(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("center", Vec3), ("radius", C.c_float),
                ("sdf_kind", C.c_uint32), ("iterations", C.c_uint32), ("box_side", C.c_float),
                ("min_radius", C.c_float), ("fixed_radius", C.c_float), ("scale", C.c_float),
                ("sdf_radius", C.c_float), ("animated", C.c_uint32), ("center_vel", Vec3), ("scale_vel", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", Vec3), ("b", Vec3), ("exponent", C.c_float)]


class Light(C.Structure):
    _fields_ = [("pos", Vec3), ("rad", C.c_float), ("emission", Vec3), ("_pad", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("res_w", C.c_float), ("res_h", C.c_float), ("vfov_or_size", C.c_float),
                ("origin", Vec3), ("at", Vec3), ("up", Vec3), ("aperture", C.c_float), ("focus", Vec3),
                ("animated", C.c_uint32), ("origin_vel", Vec3), ("at_vel", Vec3), ("up_vel", Vec3), ("focus_vel", Vec3)]


class WorldDesc(C.Structure):
    _fields_ = [("n_hitables", C.c_uint32), ("n_materials", C.c_uint32), ("n_lights", C.c_uint32),
                ("hitables", Hitable * MAX_HITABLES), ("materials", Material * MAX_MATERIALS),
                ("lights", Light * MAX_LIGHTS), ("camera", Camera),
                ("has_scattering", C.c_uint32), ("coeff_scattering", C.c_float),
                ("has_extinction", C.c_uint32), ("coeff_extinction", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("max_bounces", C.c_uint32),
                ("volume_marches", C.c_uint32), ("frame", C.c_uint32),
                ("time_start", C.c_float), ("time_end", C.c_float),
                ("max_marches", C.c_uint32), ("max_vis_marches", C.c_uint32),
                ("sdf_detail_scale", C.c_float), ("world_radius", C.c_float),
                ("tile_first", C.c_uint32), ("tile_step", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("shaded_slots", C.c_uint64),
                ("tiles", C.c_uint64), ("batches", C.c_uint64), ("ms_total", C.c_double),
                ("ms_raygen", C.c_double), ("ms_extend", C.c_double), ("ms_bin", C.c_double),
                ("ms_shade", C.c_double), ("ms_compact", C.c_double), ("ms_resolve", C.c_double),
                ("launches_extend", C.c_uint64), ("launches_shade", C.c_uint64), ("queue_bytes_bin", C.c_uint64),
                ("ms_shadow", C.c_double), ("ms_finish", C.c_double), ("queue_bytes_compact", C.c_uint64), ("shadow_jobs", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}
