"""ctypes binding of libsosgpu.so (C ABI declared in include/sosgpu.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this module
raises at first use, loudly.  Build it with `python -m <pkg>.build` / __graft_entry__.build().
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SOSGPU_LIB selects an alternative build of the same library (A/B experiments); default = in-tree libsosgpu.so
SO_PATH = os.environ.get("SOSGPU_LIB") or os.path.join(HERE, "libsosgpu.so")

# every symbol include/sosgpu.h declares (checked by tests/test_cabi.py)
EXPORTS = [
    "sosgpu_strerror", "sosgpu_last_hip_error", "sosgpu_device_count", "sosgpu_version",
    "sosgpu_create", "sosgpu_destroy", "sosgpu_set_surface_matrices", "sosgpu_set_surface_matrices_async", "sosgpu_noyaux",
    "sosgpu_noyaux_fetch", "sosgpu_os_solve", "sosgpu_aggregate", "sosgpu_ctx_bytes",
    "sosgpu_os_flops", "sosgpu_last_solve_ms", "sosgpu_profile", "sosgpu_profile_nogas", "sosgpu_glitter", "sosgpu_mat_fresnel_host", "sosgpu_trphi",
    "sosgpu_debug_phase_buffer", "sosgpu_debug_scratch", "sosgpu_comm_unique_id", "sosgpu_comm_init_rank", "sosgpu_comm_destroy",
    "sosgpu_pack", "sosgpu_unpack", "sosgpu_reduce", "sosgpu_absprofile", "sosgpu_land_surface", "sosgpu_mie", "sosgpu_granu",
    "sosgpu_granu_batch",
    "sosgpu_ctx_table_entry_bytes", "sosgpu_ctx_table", "sosgpu_os_solve_multi", "sosgpu_trim",
]
NOGAS_LEVELS = 608     # SOSGPU_NOGAS_LEVELS
SCAL_BASE = 10          # SOSGPU_SCAL_BASE: scalar block of sosgpu_aggregate = SCAL_BASE + N doubles


class GranuJob(C.Structure):
    """sosgpu_granu_job (include/sosgpu.h)."""
    _fields_ = [("d_rec", C.c_void_p), ("nalpha", C.c_int32), ("igranu", C.c_int32), ("v1", C.c_double), ("v2", C.c_double),
                ("v3", C.c_double), ("wa", C.c_double), ("alphaf", C.c_double)]


class SosgpuError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().sosgpu_strerror(code).decode()
        hip = lib().sosgpu_last_hip_error()
        super().__init__("%s failed: %s (code %d, hip error %d)" % (where, msg, code, hip))


class Wave(C.Structure):
    """struct sosgpu_wave"""
    _fields_ = [("n", C.c_int32), ("os_nb", C.c_int32), ("n0", C.c_int32), ("imat_surf", C.c_int32),
                ("ifresnel", C.c_int32), ("ipolar", C.c_int32), ("igmax", C.c_int32), ("reserved", C.c_int32),
                ("ro", C.c_double), ("ind_surf", C.c_double), ("ron", C.c_double)]


class Land(C.Structure):
    """struct sosgpu_land"""
    _fields_ = [("isurf", C.c_int32), ("reserved", C.c_int32), ("k0", C.c_double), ("k1", C.c_double), ("k2", C.c_double),
                ("alpha", C.c_double), ("beta", C.c_double), ("coef_c", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libsosgpu.so not found at %s: the HIP extension is required (no CPU fallback). "
                "Run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950)." % SO_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 and this library links against the system's.  Loaded
        # first, the system runtime would be initialised here and torch's copy afterwards -- two runtimes, and the later one
        # sees no device.  Importing torch first makes both resolve to the runtime torch loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
        L.sosgpu_strerror.restype = C.c_char_p
        L.sosgpu_strerror.argtypes = [i32]
        L.sosgpu_version.restype = C.c_char_p
        L.sosgpu_last_hip_error.restype = i32
        L.sosgpu_device_count.restype = i32
        L.sosgpu_create.restype = i32
        L.sosgpu_create.argtypes = [C.POINTER(vp), i32, C.POINTER(Wave), vp, vp, vp, vp, vp, vp, i32]
        L.sosgpu_destroy.restype = i32
        L.sosgpu_destroy.argtypes = [vp]
        L.sosgpu_set_surface_matrices.restype = i32
        L.sosgpu_set_surface_matrices.argtypes = [vp, vp]
        L.sosgpu_set_surface_matrices_async.restype = i32
        L.sosgpu_set_surface_matrices_async.argtypes = [vp, vp, vp]
        L.sosgpu_noyaux.restype = i32
        L.sosgpu_noyaux.argtypes = [vp, vp]
        L.sosgpu_noyaux_fetch.restype = i32
        L.sosgpu_noyaux_fetch.argtypes = [vp, i32, vp]
        L.sosgpu_os_solve.restype = i32
        L.sosgpu_os_solve.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sosgpu_trim.restype = i32
        L.sosgpu_trim.argtypes = []
        L.sosgpu_ctx_table_entry_bytes.restype = C.c_size_t
        L.sosgpu_ctx_table_entry_bytes.argtypes = []
        L.sosgpu_ctx_table.restype = i32
        L.sosgpu_ctx_table.argtypes = [C.POINTER(vp), i32, vp, vp]
        L.sosgpu_os_solve_multi.restype = i32
        L.sosgpu_os_solve_multi.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sosgpu_aggregate.restype = i32
        L.sosgpu_aggregate.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sosgpu_comm_unique_id.restype = i32
        L.sosgpu_comm_unique_id.argtypes = [vp]
        L.sosgpu_comm_init_rank.restype = i32
        L.sosgpu_comm_init_rank.argtypes = [C.POINTER(vp), i32, vp, i32]
        L.sosgpu_comm_destroy.restype = i32
        L.sosgpu_comm_destroy.argtypes = [vp]
        L.sosgpu_pack.restype = i32
        L.sosgpu_pack.argtypes = [vp, i32, vp, vp, vp, vp]
        L.sosgpu_unpack.restype = i32
        L.sosgpu_unpack.argtypes = [vp, i32, vp, vp, vp, vp]
        L.sosgpu_absprofile.restype = i32
        L.sosgpu_absprofile.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.sosgpu_mie.restype = i32
        L.sosgpu_mie.argtypes = [i32, i32, vp, dbl, dbl, i32, vp, vp, vp, vp]
        L.sosgpu_granu.restype = i32
        L.sosgpu_granu.argtypes = [i32, i32, i32, vp, i32, dbl, dbl, dbl, dbl, dbl, vp, vp]
        L.sosgpu_granu_batch.restype = i32
        L.sosgpu_granu_batch.argtypes = [i32, i32, i32, C.POINTER(GranuJob), vp, vp, C.c_size_t, vp]
        L.sosgpu_reduce.restype = i32
        L.sosgpu_reduce.argtypes = [vp, vp, i32, vp, vp]
        L.sosgpu_ctx_bytes.restype = C.c_size_t
        L.sosgpu_ctx_bytes.argtypes = [vp]
        L.sosgpu_profile.restype = i32
        L.sosgpu_profile.argtypes = [vp, i32, dbl, dbl, dbl, dbl, i32, i32, vp, vp, dbl, dbl, dbl, dbl, i32,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sosgpu_profile_nogas.restype = i32
        L.sosgpu_profile_nogas.argtypes = [i32, dbl, dbl, dbl, dbl, vp, vp]
        L.sosgpu_os_flops.restype = i32
        L.sosgpu_os_flops.argtypes = [vp, i32, vp, vp, vp, C.POINTER(dbl)]
        L.sosgpu_last_solve_ms.restype = i32
        L.sosgpu_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.sosgpu_glitter.restype = i32
        L.sosgpu_glitter.argtypes = [i32, i32, vp, vp, dbl, dbl, i32, i32, i32, vp, vp, vp, vp]
        L.sosgpu_mat_fresnel_host.restype = i32
        L.sosgpu_mat_fresnel_host.argtypes = [i32, vp, vp, dbl, i32, vp]
        L.sosgpu_trphi.restype = i32
        L.sosgpu_trphi.argtypes = [vp, i32, vp, dbl, dbl, i32, vp, i32, dbl, C.POINTER(Land), vp, vp]
        L.sosgpu_land_surface.restype = i32
        L.sosgpu_land_surface.argtypes = [i32, C.POINTER(Land), i32, vp, vp, dbl, i32, i32, i32, vp, C.POINTER(C.c_int32), vp]
        L.sosgpu_debug_phase_buffer.restype = i32
        L.sosgpu_debug_phase_buffer.argtypes = [vp, vp]
        L.sosgpu_debug_scratch.restype = i32
        L.sosgpu_debug_scratch.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def check(code, where):
    if code != 0:
        raise SosgpuError(code, where)
