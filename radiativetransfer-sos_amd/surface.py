"""Sea-surface reflection matrices (host side of sosgpu_glitter).

Mirrors the reference's SOS_SURFACE dispatcher for ISURF=1 (src/SOS_SURFACE.F:341, glitter branch
-> SOS_GLITTER src/SOS_GLITTER.F:229) without its file cache: the matrices are produced on the GPU and stay
in HBM for SosContext(imat_surf=1, rsurf=...).
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def mat_fresnel(mu, chr_, ind, os_ns):
    """SOS_MAT_FRESNEL (SOS_SURFACE.F:1235-1603) incl. its 4(E15.8) text round trip: host routine of the
    C ABI (sosgpu_mat_fresnel_host).  Returns alpha, beta, gamma, zeta as a (4, os_ns+1) array."""
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    chr_ = np.ascontiguousarray(chr_, dtype=np.float64)
    out = np.zeros((4, os_ns + 1))
    capi.check(capi.lib().sosgpu_mat_fresnel_host(len(mu), mu.ctypes.data_as(C.c_void_p), chr_.ctypes.data_as(C.c_void_p),
                                                  float(ind), int(os_ns), out.ctypes.data_as(C.c_void_p)),
               "sosgpu_mat_fresnel_host")
    return out


def glitter_matrices(mu, chr_, wind, ind, os_nb, os_ns=None, os_nm=None, device=0):
    """Cox-Munk reflection matrices for wind speed `wind` (m/s) and water index `ind`.
    os_ns defaults to 2*(number of Gauss angles) like SOS_ANGLES.F:325 would for these angles is NOT assumed:
    pass it explicitly when mirroring a reference run; default os_ns = os_nb, os_nm = os_nb + os_ns.
    Returns dict(rsurf=float32 cuda [os_nb+1][9][N][N], il=int32 cuda [npairs], e=float64 cuda [npairs][os_nm+1])."""
    if not torch.cuda.is_available():
        raise RuntimeError("glitter_matrices needs a GPU (gfx950); there is no CPU fallback in the product path")
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    chr_ = np.ascontiguousarray(chr_, dtype=np.float64)
    n = len(mu)
    os_ns = os_nb if os_ns is None else int(os_ns)
    os_nm = os_nb + os_ns if os_nm is None else int(os_nm)
    dev = torch.device("cuda", device)
    npairs = n * (n + 1) // 2
    rsurf = torch.zeros((os_nb + 1, 9, n, n), dtype=torch.float32, device=dev)
    il = torch.zeros(npairs, dtype=torch.int32, device=dev)
    e = torch.zeros((npairs, os_nm + 1), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    capi.check(capi.lib().sosgpu_glitter(device, n, mu.ctypes.data_as(C.c_void_p), chr_.ctypes.data_as(C.c_void_p),
                                         float(wind), float(ind), int(os_nb), os_ns, os_nm, C.c_void_p(rsurf.data_ptr()),
                                         C.c_void_p(il.data_ptr()), C.c_void_p(e.data_ptr()), st), "sosgpu_glitter")
    return dict(rsurf=rsurf, il=il, e=e)


def land_model(isurf, k0, k1, k2, alpha=0.0, beta=0.0, coef_c=0.0):
    """capi.Land of -SURF.Type isurf (3 Roujean, 4 + Rondeaux-Herman, 5 + Breon, 6 + Nadal, 7 + Maignan)."""
    return capi.Land(isurf=int(isurf), reserved=0, k0=float(k0), k1=float(k1), k2=float(k2), alpha=float(alpha), beta=float(beta),
                     coef_c=float(coef_c))


def land_matrices(land, mu, chr_, ind, os_nb, os_ns, os_nm, device=0):
    """Land-surface reflection matrices (SOS_ROUJEAN / SOS_SURFACE_BPDF / SOS_BPDF_AJOUT_BRDF) on the GPU:
    float32 cuda tensor [os_nb+1][9][N][N] in surface-file record order, for SosContext(imat_surf=1, rsurf=...).
    Raises ValueError where the reference returns IER = -1 (negative Roujean BRDF)."""
    if not torch.cuda.is_available():
        raise RuntimeError("land_matrices needs a GPU (gfx950); there is no CPU fallback in the product path")
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    chr_ = np.ascontiguousarray(chr_, dtype=np.float64)
    n = len(mu)
    dev = torch.device("cuda", device)
    rsurf = torch.zeros((os_nb + 1, 9, n, n), dtype=torch.float32, device=dev)
    ier = C.c_int32(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    capi.check(capi.lib().sosgpu_land_surface(device, C.byref(land), n, mu.ctypes.data_as(C.c_void_p),
                                              chr_.ctypes.data_as(C.c_void_p), float(ind), int(os_nb), int(os_ns), int(os_nm),
                                              C.c_void_p(rsurf.data_ptr()), C.byref(ier), st), "sosgpu_land_surface")
    if ier.value != 0:
        raise ValueError("SOS_FSF_ROUJEAN: BRDF < 0 for some geometry -- unsuitable Roujean coefficients (IER = -1)")
    return rsurf
