"""ctypes wrapper of oracle/libsos_oracle.so (the plain-C restatement, oracle/sos_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libsos_oracle.so")
_lib = None


def build():
    """Compile the C restatement (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.sos_oracle_os.restype = C.c_int
        _lib.sos_oracle_profile_rescale.restype = C.c_int
        _lib.sos_oracle_aggregate.restype = C.c_int
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)


def noyaux(is_, rmu0, mu, os_nb, alpha, beta, gamma, zeta):
    n = len(mu)
    w = 2 * n + 1
    keep = []

    def inp(x):
        a, p = _d(x)
        keep.append(a)
        return p

    vec = [np.zeros(w) for _ in range(3)]
    mats = [np.zeros((w, w)) for _ in range(6)]
    lib().sos_oracle_noyaux(C.c_int(is_), C.c_int(n), C.c_double(rmu0), inp(mu), C.c_int(os_nb),
                            inp(alpha), inp(beta), inp(gamma), inp(zeta),
                            *[v.ctypes.data_as(C.c_void_p) for v in vec],
                            *[m.ctypes.data_as(C.c_void_p) for m in mats])
    out = dict(zip(["BP", "GR", "GT", "ARR", "ART", "ATT"], mats))
    out.update(dict(zip(["XPL", "XRL", "XTL"], vec)))
    return out


def sos_os(rmu, ga, os_nb, h, xdel, ydel, alpha, beta, gamma, zeta, *, n0, tetas=0.0, ro=0.0,
           imat_surf=0, ifresnel=0, ind_surf=1.34, zprof=None, ron=float(np.float32(0.0279)), zout=-1.0,
           igmax=100, iborm=None, ipolar=1, rsurf=None):
    """Same signature/return as oracle.ref_ctypes.sos_os (minus the log)."""
    n = len(rmu)
    nt = len(h) - 1
    w = 2 * n + 1
    if iborm is None:
        iborm = os_nb
    if zprof is None:
        zprof = np.linspace(120.0, 0.0, nt + 1)
    keep = []

    def inp(x):
        a, p = _d(x)
        keep.append(a)
        return p

    rec = np.zeros((iborm + 1, 3, w))
    n_orders = C.c_int(0)
    ig_last = np.zeros(iborm + 1, dtype=np.int32)
    emoins, eplus = C.c_double(0), C.c_double(0)
    rs_p = None
    if imat_surf == 1:
        rs = np.ascontiguousarray(rsurf, dtype=np.float32)
        assert rs.shape == (iborm + 1, 9, n, n), rs.shape
        keep.append(rs)
        rs_p = rs.ctypes.data_as(C.c_void_p)
    ier = lib().sos_oracle_os(
        C.c_int(n), inp(rmu), inp(ga), C.c_int(os_nb), C.c_int(nt), C.c_int(n0), C.c_double(tetas),
        C.c_double(ro), C.c_int(imat_surf), C.c_int(ifresnel), C.c_double(ind_surf),
        inp(h), inp(xdel), inp(ydel), inp(zprof), C.c_double(ron),
        inp(alpha), inp(beta), inp(gamma), inp(zeta), C.c_double(zout), C.c_int(igmax), C.c_int(iborm),
        C.c_int(ipolar), rs_p, rec.ctypes.data_as(C.c_void_p), C.byref(n_orders),
        ig_last.ctypes.data_as(C.c_void_p), C.byref(emoins), C.byref(eplus))
    f = n_orders.value
    return dict(records=rec[:f].copy(), emoins=emoins.value, eplus=eplus.value, ier=ier,
                ig_counts=ig_last[:f].copy())


def profile_rescale(h, xdel, ydel, a_tronc, piz, piztr, os_nb):
    h = np.array(h, dtype=np.float64)
    xdel = np.array(xdel, dtype=np.float64)
    ydel = np.array(ydel, dtype=np.float64)
    iborm = lib().sos_oracle_profile_rescale(C.c_int(len(h) - 1), C.c_double(a_tronc), C.c_double(piz),
                                             C.c_double(piztr), C.c_int(os_nb),
                                             h.ctypes.data_as(C.c_void_p), xdel.ctypes.data_as(C.c_void_p),
                                             ydel.ctypes.data_as(C.c_void_p))
    return h, xdel, ydel, iborm


def aggregate(rec_bins, nf, aik, scal_bins):
    """rec_bins [nb][fmax][3][W], nf [nb], aik [nb], scal_bins [nb][7] -> (out_rec[F][3][W], out_scal[7])."""
    rec_bins = np.ascontiguousarray(rec_bins, dtype=np.float64)
    nb, fmax, _, w = rec_bins.shape
    nf = np.ascontiguousarray(nf, dtype=np.int32)
    aik = np.ascontiguousarray(aik, dtype=np.float64)
    scal_bins = np.ascontiguousarray(scal_bins, dtype=np.float64)
    out_rec = np.zeros((fmax, 3, w))
    out_scal = np.zeros(7)
    f = lib().sos_oracle_aggregate(C.c_int(nb), C.c_int(fmax), C.c_int(w), nf.ctypes.data_as(C.c_void_p),
                                   aik.ctypes.data_as(C.c_void_p), rec_bins.ctypes.data_as(C.c_void_p),
                                   scal_bins.ctypes.data_as(C.c_void_p), out_rec.ctypes.data_as(C.c_void_p),
                                   out_scal.ctypes.data_as(C.c_void_p))
    return out_rec[:f], out_scal
