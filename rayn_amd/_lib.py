"""Loader of the in-tree HIP library (rayn_amd/csrc/librayn_hip.so) — the ONLY compute backend.
There is no CPU fallback: if the library is missing or no GPU is present, calls fail loudly."""
import ctypes as C
import os
import subprocess

from . import _abi


class RaynHipError(RuntimeError):
    pass


_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
DEFAULT_LIB_PATH = os.path.join(_CSRC, "librayn_hip.so")


def _lib_path():
    """The in-tree product build.  RAYN_HIP_LIB names an alternative build of the same library (timing experiments: `make variant`)
    and is honoured ONLY together with RAYN_HIP_ALLOW_VARIANT=1 - an experiment build can never be picked up silently; the
    library reports its variant name (rayn_hip_build_variant, "" = product) and bench.py prints it."""
    alt = os.environ.get("RAYN_HIP_LIB")
    if not alt:
        return DEFAULT_LIB_PATH
    if os.environ.get("RAYN_HIP_ALLOW_VARIANT") != "1":
        raise RaynHipError(f"RAYN_HIP_LIB={alt} is set but RAYN_HIP_ALLOW_VARIANT=1 is not: refusing to load a non-product build of librayn_hip.so")
    return alt



# every symbol include/rayn_hip.h declares
EXPORTS = [
    "rayn_hip_create", "rayn_hip_create_multi", "rayn_hip_device_count", "rayn_hip_table_broadcasts", "rayn_hip_destroy", "rayn_hip_last_error", "rayn_hip_upload_world", "rayn_hip_render_frame",
    "rayn_hip_render_frame_device", "rayn_hip_get_stats", "rayn_sets_1d", "rayn_sets_2d", "rayn_build_rd_tables",
    "rayn_build_scramble", "rayn_build_fis_table", "rayn_build_fis_table_ex", "rayn_tile_count", "rayn_hip_set_profiling", "rayn_hip_get_eval_counts",
    "rayn_hip_set_batch_paths", "rayn_hip_set_cold_bytes", "rayn_hip_set_workers", "rayn_hip_set_tile_subset", "rayn_hip_set_trace_tile", "rayn_hip_get_trace", "rayn_hip_fma_policy", "rayn_hip_set_fma_policy", "rayn_hip_sizeof", "rayn_hip_probe_sdf_dist",
    "rayn_hip_probe_extend", "rayn_hip_probe_shadow", "rayn_hip_probe_detmath", "rayn_hip_build_variant",
    "rayn_hip_get_entry_stats", "rayn_hip_get_sdf_iterations", "rayn_hip_get_elision_counts", "rayn_hip_get_stage_slots", "rayn_share_pixels", "rayn_hip_render_frame_packed_device", "rayn_hip_unpack_share_device",
]


def build(force=False):
    """Compile the HIP extension for gfx950 with hipcc (rayn_amd/csrc/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        LIB_PATH = _lib_path()
        if not os.path.exists(LIB_PATH):
            raise RaynHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # PyTorch bundles its own libamdhip64.so.7; the loader keeps ONE copy per soname, and torch only works with
        # its own.  Load torch's runtime first so that librayn_hip.so binds to the same one (plain C/C++ hosts that
        # never import torch use /opt/rocm's).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        fp, up, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
        L.rayn_hip_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rayn_hip_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.rayn_hip_device_count.argtypes = [vp]
        L.rayn_hip_table_broadcasts.argtypes = [vp]
        L.rayn_hip_table_broadcasts.restype = C.c_uint64
        L.rayn_hip_destroy.argtypes = [vp]
        L.rayn_hip_destroy.restype = None
        L.rayn_hip_last_error.argtypes = [vp]
        L.rayn_hip_last_error.restype = C.c_char_p
        L.rayn_hip_upload_world.argtypes = [vp, C.POINTER(_abi.WorldDesc)]
        L.rayn_hip_render_frame.argtypes = [vp, C.POINTER(_abi.FrameParams), fp, fp, fp, fp, fp, fp, fp, fp]
        L.rayn_hip_render_frame_device.argtypes = [vp, C.POINTER(_abi.FrameParams)] + [vp] * 9
        L.rayn_hip_get_stats.argtypes = [vp, C.POINTER(_abi.Stats)]
        L.rayn_hip_get_entry_stats.argtypes = [vp, C.c_int, C.POINTER(_abi.Stats)]
        L.rayn_share_pixels.restype = C.c_uint64
        L.rayn_share_pixels.argtypes = [C.POINTER(_abi.FrameParams)]
        L.rayn_hip_render_frame_packed_device.argtypes = [vp, C.POINTER(_abi.FrameParams)] + [vp] * 6
        L.rayn_hip_unpack_share_device.argtypes = [vp, C.POINTER(_abi.FrameParams)] + [vp] * 6
        L.rayn_sets_1d.restype = C.c_uint32
        L.rayn_sets_1d.argtypes = [C.c_uint32, C.c_uint32]
        L.rayn_sets_2d.restype = C.c_uint32
        L.rayn_sets_2d.argtypes = [C.c_uint32, C.c_uint32]
        L.rayn_build_rd_tables.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, fp, fp]
        L.rayn_build_scramble.argtypes = [C.c_uint32, C.c_uint32, fp]
        L.rayn_build_fis_table.argtypes = [C.c_uint32, C.c_float, fp]
        L.rayn_build_fis_table_ex.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float, fp]
        L.rayn_tile_count.restype = C.c_uint32
        L.rayn_tile_count.argtypes = [C.c_uint32] * 4
        L.rayn_hip_set_profiling.argtypes = [vp, C.c_int, C.c_int]
        L.rayn_hip_get_eval_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rayn_hip_get_sdf_iterations.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rayn_hip_get_elision_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rayn_hip_get_stage_slots.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rayn_hip_set_batch_paths.argtypes = [vp, C.c_uint64]
        L.rayn_hip_set_cold_bytes.argtypes = [vp, C.c_uint64]
        L.rayn_hip_set_workers.argtypes = [vp, C.c_int, C.c_uint64]
        L.rayn_hip_set_tile_subset.argtypes = [vp, up, C.c_uint32]
        L.rayn_hip_set_trace_tile.argtypes = [vp, C.c_int]
        L.rayn_hip_get_trace.argtypes = [vp, up, C.c_uint64]
        L.rayn_hip_get_trace.restype = C.c_int64
        L.rayn_hip_set_fma_policy.argtypes = [vp, C.c_int]
        L.rayn_hip_sizeof.restype = C.c_size_t
        L.rayn_hip_sizeof.argtypes = [C.c_int]
        L.rayn_hip_probe_sdf_dist.argtypes = [vp, C.POINTER(_abi.FrameParams), C.c_uint32, fp, fp, C.c_uint32]
        L.rayn_hip_probe_extend.argtypes = [vp, C.POINTER(_abi.FrameParams), C.c_uint32, fp, fp, fp, up, C.c_uint32]
        L.rayn_hip_probe_shadow.argtypes = [vp, C.POINTER(_abi.FrameParams), fp, fp, fp, C.c_uint32]
        L.rayn_hip_probe_detmath.argtypes = [vp, C.c_uint32, fp, fp, fp, C.c_uint32]
        L.rayn_hip_build_variant.restype = C.c_char_p
        L.rayn_hip_build_variant.argtypes = []
        _lib = L
    return _lib


def build_variant():
    """"" for the product build, else the name the library was built with (`make variant VARIANT=...`)."""
    return lib().rayn_hip_build_variant().decode()
