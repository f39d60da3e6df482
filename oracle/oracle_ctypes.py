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


def stop_margin():
    """Tie audit of the last sos_os call: min |Z1/threshold - 1| over all its stop decisions (sos_oracle.c audit)."""
    lib().sos_oracle_stop_margin.restype = C.c_double
    return lib().sos_oracle_stop_margin()


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


def sigma2(wind):
    lib().sos_oracle_sigma2.restype = C.c_double
    return lib().sos_oracle_sigma2(C.c_double(wind))


def glitter(rmu, chr_, wind, ind, os_nb, os_ns, os_nm):
    """SOS_GLITTER restatement: returns dict(rsurf float32[os_nb+1][9][N][N], il[npairs], e[npairs][os_nm+1],
    coef[4][os_ns+1])."""
    n = len(rmu)
    npairs = n * (n + 1) // 2
    a_mu, p_mu = _d(rmu)
    a_ch, p_ch = _d(chr_)
    out = np.zeros((os_nb + 1, 9, n, n), dtype=np.float32)
    il = np.zeros(npairs, dtype=np.int32)
    e = np.zeros((npairs, os_nm + 1))
    coef = np.zeros((4, os_ns + 1))
    lib().sos_oracle_glitter(C.c_int(n), p_mu, p_ch, C.c_double(wind), C.c_double(ind), C.c_int(os_nb), C.c_int(os_ns),
                             C.c_int(os_nm), out.ctypes.data_as(C.c_void_p), il.ctypes.data_as(C.c_void_p),
                             e.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p))
    return dict(rsurf=out, il=il, e=e, coef=coef)


def trphi(rmu, rec, tau, tauout, phi, *, igli=0, n0=1, wind=0.0, ind_surf=1.34, ifresnel=0, ipolar=1):
    n = len(rmu)
    w = 2 * n + 1
    a_mu, p_mu = _d(rmu)
    a_rec, p_rec = _d(rec)
    outs = [np.zeros(w) for _ in range(4)]
    lib().sos_oracle_trphi(C.c_int(n), p_mu, C.c_int(a_rec.shape[0]), p_rec, C.c_double(tau), C.c_double(tauout),
                           C.c_double(phi), C.c_int(igli), C.c_int(n0), C.c_double(wind), C.c_double(ind_surf),
                           C.c_int(ifresnel), C.c_int(ipolar), *[o.ctypes.data_as(C.c_void_p) for o in outs])
    return outs


def polar(xi, xq, xu):
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    lib().sos_oracle_polar(C.c_double(xi), C.c_double(xq), C.c_double(xu), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def sos_profile(tr, hr, ta, ha, altabs=None, tabs=None, absprofil=1):
    """SOS_PROFILE for IPROFIL=1 (SOS_PROFIL.F:224) as read back from the PROFIL file: returns dict(ier, nt, zprof, h,
    xdel (=PCAER), ydel (=PCMOL)).  tabs=None or all-zero last level: no gas absorption."""
    n = 601
    z, h, pa, pm = (np.zeros(n) for _ in range(4))
    nt = C.c_int(0)
    if tabs is None:
        p_alt, p_tab, absprofil = None, None, 7
    else:
        a_alt = np.ascontiguousarray(altabs, dtype=np.float64); a_tab = np.ascontiguousarray(tabs, dtype=np.float64)
        assert a_alt.shape == (50,) and a_tab.shape == (50,)
        p_alt, p_tab = a_alt.ctypes.data_as(C.c_void_p), a_tab.ctypes.data_as(C.c_void_p)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().sos_profile_oracle.restype = C.c_int
    ier = lib().sos_profile_oracle(C.c_double(tr), C.c_double(hr), C.c_double(ta), C.c_double(ha), C.c_int(absprofil),
                                   p_alt, p_tab, C.byref(nt), vp(z), vp(h), vp(pa), vp(pm))
    k = nt.value + 1
    return dict(ier=ier, nt=nt.value, zprof=z[:k].copy(), h=h[:k].copy(), xdel=pa[:k].copy(), ydel=pm[:k].copy())
