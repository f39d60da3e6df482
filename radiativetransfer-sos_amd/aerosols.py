"""Aerosol optical properties of one wavelength: what the reference's SOS_AEROSOLS writes to Aerosols.txt
for every `-AER.Model`: 0 mono-modal log-normal / Junge, 1 WMO, 2 Shettle & Fenn, 3 bimodal log-normal, 4 external phase
functions, 5 user mixture of log-normal / Junge modes.

    mie_angles        <- SOS_ANGLES for the Mie angle set  src/SOS_ANGLES.F:380-466 (Gauss nodes, D21.14 text values)
    alpha_grid        <- the size-parameter loop of SOS_MIE  src/SOS_MIE.F:434-443, 707-708
    mie_records       <- SOS_MIE + SOS_FPHASE_MIE on the GPU (csrc/mie.hip, C ABI sosgpu_mie), no MIE cache file
    size_integral     <- SOS_MIE + SOS_GRANU  src/SOS_AEROSOLS.F:4392-4820  size-distribution integral of the Mie records, on
                         the GPU as well (sosgpu_granu; the numpy restatement of SOS_GRANU is tests/aerosol_loops.py)
    decompo_legendre  <- SOS_DECOMPO_LEGENDRE  src/SOS_AEROSOLS.F:3924-4390  truncation + Legendre expansions
    init_param_wmo    <- SOS_INIT_PARAMWMO  src/SOS_AEROSOLS.F:3334   (component table $SOS_ABS_ROOT/fic/Data_WMO_*)
    init_param_sf     <- SOS_INIT_PARAMSF   src/SOS_AEROSOLS.F:3557   (Data_SF_*, IRefrac_* tables)
    aerosols          <- SOS_AEROSOLS  src/SOS_AEROSOLS.F:680 (IMOD = 0: :1150-1290; 1: :1312-1510; 2: :1517-1705;
                         3: :1710-2125; 4: :2143-2280; 5: :2289-2770; closing :2771-2890)

REAL*4 variables and literals of the Fortran are kept REAL*4 (`np.float32`) where they decide a value."""
import collections
import contextlib
import ctypes as C
import functools
import math
import os
import threading

import numpy as np

from . import capi

_F = lambda x: float(np.float32(x))
MIE_ALPHAMIN = 0.0001                    # CTE_MIE_ALPHAMIN (a D+00 literal, SOS.h:116)
COEF_NRMAX = _F(0.0001)                  # SOS.h:134
WAMIN = _F(0.364)                        # SOS.h:70
MU1_TRONCA, MU2_TRONCA = _F(0.8), _F(0.94)     # SOS.h:166-167
SEUIL_TRONCA = _F(0.1)                   # CTE_PH_SEUIL_TRONCA, SOS.h:172


class AerosolError(RuntimeError):
    """IER = -1 of SOS_AEROSOLS and its leaves."""


def _round_sig(x, sig):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    nz = x != 0
    mag = np.floor(np.log10(np.abs(x[nz])))
    scale = 10.0 ** (sig - 1 - mag)
    out[nz] = np.round(x[nz] * scale) / scale
    return out


@functools.lru_cache(maxsize=16)
def mie_angles(nb_gauss):
    """Mie angle set without user angles: positive Gauss nodes of the 2 nb_gauss-point rule, ascending, as re-read from
    Aer_UsedAngles.txt (D21.14).  Returns xmu[-N:N], xhr[-N:N] as arrays of 2N+1 (index j + N; entry N unused = 0)."""
    from .run_sos import sos_gauss
    x, w = sos_gauss(nb_gauss)                             # the reference's own Gauss-Legendre routine (SOS_ANGLES.F:1022)
    mu, wt = _round_sig(x, 14), _round_sig(w, 14)
    n = nb_gauss
    xmu, xhr = np.zeros(2 * n + 1), np.zeros(2 * n + 1)
    xmu[n + 1:], xhr[n + 1:] = mu, wt
    xmu[:n], xhr[:n] = -mu[::-1], wt[::-1]
    return xmu, xhr


@functools.lru_cache(maxsize=64)
def alpha_grid(alphao, alphaf):
    """Size parameters SOS_MIE steps through (ALPHA = ALPHA + PAS with the REAL*4 step literals, SOS_MIE.F:437-443,707)."""
    out = []
    a = float(alphao)
    steps = [(_F(100.), _F(1.00)), (_F(30.), _F(0.10)), (_F(10.), _F(0.05)), (_F(1.00), _F(0.01)), (_F(0.1), _F(0.001))]
    while True:
        out.append(a)
        pas = _F(0.0001)
        for lim, st in steps:
            if a > lim:
                pas = st
                break
        a = a + pas
        if not a <= alphaf:
            break
    return np.array(out)


# Device-resident Mie records of the last few (refractive index, angle set, size-parameter range) combinations: what the
# reference keeps as MIE cache files named after exactly these quantities (SOS_NOM_FICMIE, SOS_AEROSOLS.F:3128 --
# "MIE1.450-0.00300-0.0001-00100.00-MU12") and re-reads for every wavelength that shares them.  2 MB each at 40 Mie angles.
_MIE_CACHE = collections.OrderedDict()
_MIE_LOCK = threading.Lock()
_MIE_CACHE_MAX = 12


def _mie_device_records(xmu, rn, in_, alphao, alphaf, device=0):
    """(rec float32 [na][4 + 3 W], g float64 [na]) device tensors of sosgpu_mie for the grid alpha_grid(alphao, alphaf)."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("Mie theory needs a GPU (gfx950); there is no CPU fallback in the product path")
    xmu = np.ascontiguousarray(xmu, dtype=np.float64)
    key = (xmu.tobytes(), float(rn), float(in_), float(alphao), float(alphaf), int(device))
    with _MIE_LOCK:
        hit = _MIE_CACHE.get(key)
        if hit is not None:
            _MIE_CACHE.move_to_end(key)
            return hit
    al = alpha_grid(alphao, alphaf)
    w = len(xmu)
    dev = torch.device("cuda", device)
    rec = torch.zeros((len(al), 4 + 3 * w), dtype=torch.float32, device=dev)
    g = torch.zeros(len(al), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = capi.lib().sosgpu_mie(device, (w - 1) // 2, xmu.ctypes.data_as(C.c_void_p), float(rn), float(in_), len(al),
                               al.ctypes.data_as(C.c_void_p), C.c_void_p(rec.data_ptr()), C.c_void_p(g.data_ptr()), st)
    if rc == -3:
        raise AerosolError("size parameter up to %g: more Mie coefficients than the device kernel holds" % alphaf)
    capi.check(rc, "sosgpu_mie")               # (synchronous: the records are complete, usable from any stream)
    with _MIE_LOCK:
        _MIE_CACHE[key] = (rec, g)
        while len(_MIE_CACHE) > _MIE_CACHE_MAX:
            _MIE_CACHE.popitem(last=False)
    return rec, g


def mie_records(xmu, rn, in_, alphao, alphaf, device=0):
    """The records of the reference's MIE file for (rn, in_) on the grid alpha_grid(alphao, alphaf), on the host (parity
    accessor): dict(alpha, qext, qsca float32 [na]; g float64 [na]; imie, qmie, umie float32 [na][2N+1])."""
    rec, g = _mie_device_records(xmu, rn, in_, alphao, alphaf, device)
    w = len(xmu)
    r = rec.cpu().numpy()
    return dict(alpha=r[:, 0].copy(), qext=r[:, 1].copy(), qsca=r[:, 2].copy(), g=g.cpu().numpy(), imie=r[:, 4:4 + w].copy(),
                qmie=r[:, 4 + w:4 + 2 * w].copy(), umie=r[:, 4 + 2 * w:4 + 3 * w].copy(), alphaf=float(alphaf))


# ---- size integrals queued ahead of their use (run_sos.sos_spectrum) ------------------------------------------------------
# A spectrum needs one or two size integrals per wavelength, each a 0.1 ms device job whose result the host's Legendre
# expansion waits for.  sos_spectrum therefore runs the aerosol model of a batch of wavelengths once in COLLECT mode (the model
# code below records the arguments of its size_integral calls and stops before the expansion), queues all of them with
# sosgpu_granu_batch -- one workgroup per integral, results copied to pinned host memory behind them -- and the real pass finds
# its integrals ready: one wait per batch instead of one per wavelength.  State is per thread.
_TLS = threading.local()


def _granu_key(xmu, rn, in_, alphaf, igranu, v1, v2, v3, wa, device):
    return (np.ascontiguousarray(xmu, dtype=np.float64).tobytes(), float(rn), float(in_), float(alphaf), int(igranu), float(v1),
            float(v2), float(v3), float(wa), int(device))


class _GranuBatch:
    """The results of one sosgpu_granu_batch call on their way to the host (the first reader waits for the copy)."""

    def __init__(self, host, event, keep):
        self.host, self.event, self.keep = host, event, keep

    def row(self, i):
        if self.event is not None:
            self.event.synchronize()
            self.event = self.keep = None                  # (device output, work area and the records' references)
        return self.host[i]


@contextlib.contextmanager
def collect_size_integrals():
    """COLLECT mode: size_integral records its arguments in the yielded list and returns placeholders; aerosols() returns None."""
    prev = getattr(_TLS, "collect", None)
    _TLS.collect = reqs = []
    try:
        yield reqs
    finally:
        _TLS.collect = prev


def prefetch_size_integrals(requests, batch=32):
    """Queue the size integrals `requests` (keys recorded by collect_size_integrals) on the current HIP stream; size_integral
    calls of this thread with the same arguments then take their result from the batch.  Returns the number queued."""
    import torch
    ready = getattr(_TLS, "ready", None)
    if ready is None:
        ready = _TLS.ready = {}
    todo = collections.OrderedDict()
    for k in requests:
        if k not in ready and k[4] in (1, 2):
            todo.setdefault((k[0], k[9]), collections.OrderedDict())[k] = None
    queued = 0
    for (xb, device), keys in todo.items():
        keys = list(keys)
        xmu = np.frombuffer(xb, dtype=np.float64)
        w = len(xmu)
        dev = torch.device("cuda", device)
        for c0 in range(0, len(keys), batch):
            part, recs = [], []
            for k in keys[c0:c0 + batch]:
                try:
                    recs.append(_mie_device_records(xmu, k[1], k[2], MIE_ALPHAMIN, k[3], device)[0])
                    part.append(k)
                except AerosolError:                       # (the call that owns this integral reports it)
                    pass
            if not part:
                continue
            jobs = (capi.GranuJob * len(part))()
            for j, (k, rec) in enumerate(zip(part, recs)):
                jobs[j] = capi.GranuJob(rec.data_ptr(), int(rec.shape[0]), k[4], k[5], k[6], k[7], k[8], k[3])
            stride = 3 * max(int(r.shape[0]) for r in recs) + 1
            out = torch.empty((len(part), 3 + 3 * w), dtype=torch.float64, device=dev)
            work = torch.empty((len(part), stride), dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream(dev)
            capi.check(capi.lib().sosgpu_granu_batch(device, (w - 1) // 2, len(part), jobs, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(work.data_ptr()), stride, C.c_void_p(st.cuda_stream)),
                       "sosgpu_granu_batch")
            host = torch.empty(out.shape, dtype=torch.float64, pin_memory=True)
            host.copy_(out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            b = _GranuBatch(host.numpy(), ev, (out, work, recs, host))
            for j, k in enumerate(part):
                ready[k] = (b, j)
            queued += len(part)
    return queued


def drop_prefetched_size_integrals():
    _TLS.ready = None


def size_integral(xmu, rn, in_, alphaf, igranu, v1, v2, v3, wa, device=0):
    """SOS_MIE + SOS_GRANU for one aerosol mode, both on the GPU: Mie records on the grid alpha_grid(MIE_ALPHAMIN, alphaf)
    (sosgpu_mie; kept on the device and shared by the wavelengths of a spectrum, like the reference's MIE files) integrated
    over the size distribution in record order (sosgpu_granu; igranu 1: log-normal, modal radius v1, ln-std v2; 2: Junge,
    r0 = v1, slope v2, rmax = v3).  Returns kmat1, kmat2 (per particle), somme_nr, p11, p12, p33 [2N+1]."""
    import torch
    w = len(xmu)
    coll = getattr(_TLS, "collect", None)
    ready = getattr(_TLS, "ready", None)
    if coll is not None or ready:
        key = _granu_key(xmu, rn, in_, alphaf, igranu, v1, v2, v3, wa, device)
        if coll is not None:
            coll.append(key)
            o = np.ones(w)
            return 1.0, 1.0, 1.0, o, o.copy(), o.copy()
        hit = ready.get(key)
        if hit is not None:
            out = hit[0].row(hit[1])
            return float(out[0]), float(out[1]), float(out[2]), out[3:3 + w].copy(), out[3 + w:3 + 2 * w].copy(), out[3 + 2 * w:].copy()
    if igranu not in (1, 2):
        raise AerosolError("unknown size distribution %d" % igranu)
    rec, _ = _mie_device_records(xmu, rn, in_, MIE_ALPHAMIN, alphaf, device)
    out = np.zeros(3 + 3 * w)
    st = C.c_void_p(torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream)
    capi.check(capi.lib().sosgpu_granu(device, (w - 1) // 2, int(rec.shape[0]), C.c_void_p(rec.data_ptr()), int(igranu), float(v1),
                                       float(v2), float(v3), float(wa), float(alphaf), out.ctypes.data_as(C.c_void_p), st),
               "sosgpu_granu")
    return float(out[0]), float(out[1]), float(out[2]), out[3:3 + w].copy(), out[3 + w:3 + 2 * w].copy(), out[3 + 2 * w:].copy()


def _seq_sum(a):
    """Left-to-right sum along the last axis (np.cumsum accumulates sequentially, np.sum pairwise): the order of the Fortran loops."""
    return np.cumsum(a, axis=-1)[..., -1]


@functools.lru_cache(maxsize=16)
def _legendre_tables(xmu_bytes, os_nb):
    """What SOS_DECOMPO_LEGENDRE computes from the angles and orders alone: Legendre polynomials PL(K) and the generalised
    functions POL(K) at the Mie angles (SOS_AEROSOLS.F:4100-4180) and, per order I, the REAL*4 coefficient expressions CO1, CO2
    and X2 of the alpha / zeta sums (:4330-4375, as in SOS_MAT_FRESNEL) with the indices they multiply.  Cached per angle set."""
    xmu = np.frombuffer(xmu_bytes, dtype=np.float64)
    w = len(xmu)
    n = (w - 1) // 2
    sel = np.array([j for j in range(w) if j != n])                 # J = -N..N without 0, ascending
    xr = xmu[sel]
    # Legendre polynomials P_k(xr): PL(K+1) = ((2K+1) X PL(K) - K PL(K-1)) / (K+1)
    pl = np.zeros((os_nb + 2, len(sel)))
    pl[0] = 1.
    plm = np.zeros(len(sel))
    for k in range(os_nb + 1):
        pl[k + 1] = ((2 * k + 1.) * xr * pl[k] - k * (pl[k - 1] if k else plm)) / (k + 1.)
    # generalised functions POL(K), K >= 2
    pol = np.zeros((os_nb + 2, len(sel)))
    pol[2] = 3. * (1. - xr ** 2) / 2. / math.sqrt(6.0)
    for k in range(2, os_nb + 1):
        d = (2. * k + 1.) / math.sqrt(1. * (k + 3.) * (k - 1.))
        e = math.sqrt(1. * (k + 2.) * (k - 2.)) / (2. * k + 1.)
        pol[k + 1] = d * (xr * pol[k] - e * pol[k - 1])
    f = np.float32
    # per order I = 2 .. os_nb: CO1, CO2 and the X2 weights of the four sums, as zero-padded matrices [os_nb - 1][os_nb // 2 + 1]
    # with the indices they multiply (a padded entry has weight 0 and comes LAST in its row: a sequential sum over the padded row
    # equals the Fortran loop's sum exactly, x + 0.0 = x)
    ni, wmax = os_nb - 1, os_nb // 2 + 1
    co1, co2 = np.zeros(ni), np.zeros(ni)
    xn, xm = np.zeros((ni, wmax)), np.zeros((ni, wmax))
    jn_idx, jm_idx = np.zeros((ni, wmax), dtype=np.int64), np.zeros((ni, wmax), dtype=np.int64)
    for r, i in enumerate(range(2, os_nb + 1)):               # CO1, CO2, X2 are REAL*4 expressions
        co1[r] = float(f(4) * (f(2 * i) + f(1.)) / f(i) / (f(i) - f(1.)) / (f(i) + f(1.)) / (f(i) + f(2.)))
        co2[r] = float(f(i) * (f(i) - f(1.)) / ((f(i) + f(1.)) * (f(i) + f(2.))))
        nn, mm = int(i * .5), int((i - 1) * .5)
        fi1 = (f(i) - f(1.)) * (f(i) - f(1.))
        jn = np.arange(1, nn + 1)
        xn[r, :nn] = (fi1 - f(3.) * ((2 * jn).astype(np.float32) - f(1.)) * (i - jn).astype(np.float32)).astype(np.float64)
        jn_idx[r, :nn] = i - 2 * jn
        jm = np.arange(0, mm + 1)
        xm[r, :mm + 1] = (fi1 - f(3.) * jm.astype(np.float32) * ((2 * i - 2 * jm).astype(np.float32) - f(1.))).astype(np.float64)
        jm_idx[r, :mm + 1] = i - 2 * jm - 1
    coefs = (co1, co2, xn, jn_idx, xm, jm_idx)
    for a in coefs:
        a.setflags(write=False)
    for a in (sel, xr, pl, pol):
        a.setflags(write=False)
    return sel, xr, pl, pol, coefs


def decompo_legendre(itronc, xmu, xhr, os_nb, p11_in, p12, p22, p33):
    """SOS_DECOMPO_LEGENDRE: forward-peak truncation (log-linear extrapolation of P11 beyond acos(0.94), slope taken between
    acos(0.8) and acos(0.94)) and the expansions alpha, beta, gamma, zeta (0:os_nb), normalised by beta_0.
    Returns dict(alpha, beta, gamma, zeta, beta22, delta33, coef_tronca, itronc).
    Vectorised over the angles and orders with the reference's summation order kept (sequential sums): bit-identical to
    the statement-for-statement loops of tests/aerosol_loops.py (tests/test_aerosols.py)."""
    w = len(xmu)
    n = (w - 1) // 2
    J = lambda j: j + n
    ttt = p11_in.copy()
    kk = np.arange(os_nb + 1)
    sel, xr, pl, pol, coefs = _legendre_tables(np.ascontiguousarray(xmu, dtype=np.float64).tobytes(), os_nb)
    while True:
        p11 = ttt.copy()
        if itronc:
            k1 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU1_TRONCA), None)
            k2 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU2_TRONCA), None)
            if k1 is None or k2 is None:
                raise AerosolError("truncation angles outside the Mie angle set")
            aa = (math.log10(p11[J(k2)]) - math.log10(p11[J(k1)])) / (math.acos(xmu[J(k2)]) - math.acos(xmu[J(k1)]))
            x1, x2 = math.log10(p11[J(k2)]), math.acos(xmu[J(k2)])
            for j in range(k2 + 1, n + 1):
                p11[J(j)] = 10 ** (x1 + aa * (math.acos(xmu[J(j)]) - x2))
        x = p11[sel] * xhr[sel]
        beta11 = _seq_sum(x[None, :] * pl[:os_nb + 1])
        beta11 = (2 * kk + 1) * beta11 * .5
        coef = 2 * (1 - beta11[0]) if itronc else 0.0
        if itronc and coef < SEUIL_TRONCA:
            itronc = 0                    # truncation too small to matter: start again without it (SOS_AEROSOLS.F:4195-4214)
            continue
        break
    xxx = xhr[sel] * p12[sel] * p11[sel] / ttt[sel]
    xb = xhr[sel] * p22[sel] * (p11[sel] / ttt[sel])
    xx = xhr[sel] * p33[sel] * p11[sel] / ttt[sel]
    gamma12 = np.zeros(os_nb + 1)
    gamma12[2:] = _seq_sum(xxx[None, :] * pol[2:os_nb + 1])
    beta22 = _seq_sum(xb[None, :] * pl[:os_nb + 1])
    delta33 = _seq_sum(xx[None, :] * pl[:os_nb + 1])
    beta22 = beta22 * (2. * kk + 1.) * .5
    delta33 = delta33 * (2. * kk + 1.) * .5
    gamma12 = gamma12 * (2. * kk + 1.) * .5
    # alpha_I, zeta_I for all orders at once: the four sums of every order run along the rows of the padded weight matrices,
    # sequentially (cumsum), exactly as the Fortran loops J = 1..NN / J = 0..MM add them
    co1, co2r, xn, jn_idx, xm, jm_idx = coefs
    last = lambda a: np.cumsum(a, axis=1)[:, -1]
    s1, s2 = last(xn * beta22[jn_idx]), last(xn * delta33[jn_idx])
    s3, s4 = last(xm * beta22[jm_idx]), last(xm * delta33[jm_idx])
    alp, zeta = np.zeros(os_nb + 1), np.zeros(os_nb + 1)
    co3 = co2r * delta33[2:]
    co2 = co2r * beta22[2:]
    zeta[2:] = co3 - co1 * (s2 - s3)
    alp[2:] = co2 - co1 * (s1 - s4)
    z1 = beta11[0]
    return dict(alpha=alp / z1, beta=beta11 / z1, gamma=gamma12 / z1, zeta=zeta / z1, beta22=beta22 / z1, delta33=delta33 / z1,
                coef_tronca=float(coef), itronc=itronc)


def decompo_legendre_many(itronc, xmu, xhr, os_nb, p11_in, p12, p22, p33):
    """decompo_legendre for B phase functions at once (p11_in, p12, p22, p33: [B][2N+1]; the wavelengths of a chunk of a spectrum):
    the same statements with a leading batch axis -- sums along the angle axis stay sequential, the scalar expressions of the
    truncation (log10, acos, 10 ** x of libm) are formed per phase function exactly as the single form forms them -- so every
    output equals decompo_legendre's bit for bit (tests/test_aerosols.py).  Returns a list of B dictionaries."""
    p11_in, p12, p22, p33 = (np.asarray(a, dtype=np.float64) for a in (p11_in, p12, p22, p33))
    nbatch, w = p11_in.shape
    n = (w - 1) // 2
    J = lambda j: j + n
    kk = np.arange(os_nb + 1)
    sel, xr, pl, pol, coefs = _legendre_tables(np.ascontiguousarray(xmu, dtype=np.float64).tobytes(), os_nb)
    ttt = p11_in
    p11 = ttt.copy()
    itr = np.full(nbatch, 1 if itronc else 0)
    if itronc:
        k1 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU1_TRONCA), None)
        k2 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU2_TRONCA), None)
        if k1 is None or k2 is None:
            raise AerosolError("truncation angles outside the Mie angle set")
        ac1, ac2 = math.acos(xmu[J(k1)]), math.acos(xmu[J(k2)])
        acj = [math.acos(xmu[J(j)]) for j in range(k2 + 1, n + 1)]
        for b in range(nbatch):
            x1 = math.log10(p11[b, J(k2)])
            aa = (x1 - math.log10(p11[b, J(k1)])) / (ac2 - ac1)
            for q, j in enumerate(range(k2 + 1, n + 1)):
                p11[b, J(j)] = 10 ** (x1 + aa * (acj[q] - ac2))
    wgt = pl[:os_nb + 1][None, :, :]
    beta11 = np.cumsum((p11[:, sel] * xhr[sel])[:, None, :] * wgt, axis=-1)[..., -1]
    beta11 = (2 * kk + 1) * beta11 * .5
    coef = 2 * (1 - beta11[:, 0]) if itronc else np.zeros(nbatch)
    if itronc:
        redo = coef < SEUIL_TRONCA                      # truncation too small to matter: again without it (SOS_AEROSOLS.F:4195-4214)
        if redo.any():
            p11[redo] = ttt[redo]
            b11 = np.cumsum((p11[redo][:, sel] * xhr[sel])[:, None, :] * wgt, axis=-1)[..., -1]
            beta11[redo] = (2 * kk + 1) * b11 * .5
            coef = np.where(redo, 0.0, coef)
            itr[redo] = 0
    ps, ts = p11[:, sel], ttt[:, sel]
    xxx = xhr[sel] * p12[:, sel] * ps / ts
    xb = xhr[sel] * p22[:, sel] * (ps / ts)
    xx = xhr[sel] * p33[:, sel] * ps / ts
    gamma12 = np.zeros((nbatch, os_nb + 1))
    gamma12[:, 2:] = np.cumsum(xxx[:, None, :] * pol[2:os_nb + 1][None], axis=-1)[..., -1]
    beta22 = np.cumsum(xb[:, None, :] * wgt, axis=-1)[..., -1]
    delta33 = np.cumsum(xx[:, None, :] * wgt, axis=-1)[..., -1]
    beta22 = beta22 * (2. * kk + 1.) * .5
    delta33 = delta33 * (2. * kk + 1.) * .5
    gamma12 = gamma12 * (2. * kk + 1.) * .5
    co1, co2r, xn, jn_idx, xm, jm_idx = coefs
    last = lambda a: np.cumsum(a, axis=-1)[..., -1]
    s1, s2 = last(xn * beta22[:, jn_idx]), last(xn * delta33[:, jn_idx])
    s3, s4 = last(xm * beta22[:, jm_idx]), last(xm * delta33[:, jm_idx])
    alp, zeta = np.zeros((nbatch, os_nb + 1)), np.zeros((nbatch, os_nb + 1))
    co3 = co2r * delta33[:, 2:]
    co2 = co2r * beta22[:, 2:]
    zeta[:, 2:] = co3 - co1 * (s2 - s3)
    alp[:, 2:] = co2 - co1 * (s1 - s4)
    z1 = beta11[:, :1]
    alp, bet, gam, zet, b22, d33 = alp / z1, beta11 / z1, gamma12 / z1, zeta / z1, beta22 / z1, delta33 / z1
    return [dict(alpha=alp[b], beta=bet[b], gamma=gam[b], zeta=zet[b], beta22=b22[b], delta33=d33[b], coef_tronca=float(coef[b]),
                 itronc=int(itr[b])) for b in range(nbatch)]


def _rmax_lnd(rmodal, var):
    return rmodal * math.exp(var * var) * math.exp(var * math.sqrt(-2. * math.log(COEF_NRMAX)))


def _alphaf(rmax, wa_for_grid):
    af = float(np.float32(100 + 100 * math.trunc(2. * math.pi * rmax / (100. * wa_for_grid))))
    if MIE_ALPHAMIN > af or af >= 1e5:
        raise AerosolError("size-parameter range of the Mie calculation is not valid (SOS_AEROSOLS ERROR_1009)")
    return af


def _round_index(rn, in_):
    if in_ > 0.:
        raise AerosolError("the imaginary part of the refractive index must be negative or null")
    return round(rn * 1000.) / 1000., -round(-in_ * 100000.) / 100000.


def _lnd_component(xmu, rn, in_, rmodal, var, alphaf, wa, device):
    k1, k2, _, a11, a12, a33 = size_integral(xmu, rn, in_, alphaf, 1, rmodal, var, -999.0, wa, device)
    return k1, k2, a11, a12, a33


def _mix(components, xmu, wa, device):
    """Number-weighted mixture of log-normal components [(weight, rn, in, rmodal, var, alphaf)] (SOS_AEROSOLS.F:1396-1493
    for the WMO models, :1574-1686 for Shettle & Fenn): cross sections add, phase functions add weighted by the
    scattering cross section and are normalised by the mixture's."""
    w = len(xmu)
    kmat1 = kmat2 = 0.
    p11, p12, p33 = np.zeros(w), np.zeros(w), np.zeros(w)
    for wt, rn, in_, rmodal, var, alphaf in components:
        if wt == 0.:
            continue
        if MIE_ALPHAMIN > alphaf or alphaf >= 1e5:
            raise AerosolError("size-parameter range of the Mie calculation is not valid (SOS_AEROSOLS ERROR_1009)")
        k1, k2, a11, a12, a33 = _lnd_component(xmu, rn, in_, rmodal, var, alphaf, wa, device)
        kmat1 = kmat1 + wt * k1
        kmat2 = kmat2 + wt * k2
        p11 = p11 + wt * a11 * k2
        p12 = p12 + wt * a12 * k2
        p33 = p33 + wt * a33 * k2
    return kmat1, kmat2, p11 / kmat2, p12 / kmat2, p33 / kmat2


def _fic(name):
    root = os.environ.get("SOS_ABS_ROOT", "")
    if not root:
        raise AerosolError("SOS_ABS_ROOT is not defined (SOS_AEROSOLS ERROR_925)")
    path = os.path.join(root, "fic", name)
    if not os.path.exists(path):
        raise AerosolError("aerosol data file %s not found" % path)
    return path


def _rows(path):
    with open(path) as f:
        return [[float(t) for t in ln.split()] for ln in f if ln.strip()]


def _interpol(y1, y2, x1, x2, x):
    return ((y2 - y1) / (x2 - x1)) * (x - x2) + y2          # SOS_INTERPOL, SOS_AEROSOLS.F:3844


def _round_mr_mi(mr, mi):
    # DNINT rounds half away from zero
    return math.floor(mr * 1000. + 0.5) / 1000., -math.floor(-mi * 100000. + 0.5) / 100000.


def init_param_wmo(wa):
    """SOS_INIT_PARAMWMO (SOS_AEROSOLS.F:3334-3470): modal radii, ln-variances, volumes V and the refractive indices of the
    four WMO components (dust-like, water-soluble, oceanic, soot) interpolated to wa.  A wavelength outside the table
    leaves the indices at zero, like the reference's END= branch."""
    rows = _rows(_fic("Data_WMO_cor_2015_12_16"))
    r = rows[0][:4]
    v2 = [x * math.log(10.) for x in rows[1][:4]]
    vol = rows[2][:4]
    mr, mi = [0.] * 4, [0.] * 4
    tab = rows[3:]
    for lo, hi in zip(tab[:-1], tab[1:]):
        if lo[0] <= wa <= hi[0]:
            for i in range(4):
                a = _interpol(lo[1 + 2 * i], hi[1 + 2 * i], lo[0], hi[0], wa)
                b = _interpol(lo[2 + 2 * i], hi[2 + 2 * i], lo[0], hi[0], wa)
                mr[i], mi[i] = _round_mr_mi(a, b)
            break
    return r, v2, mr, mi, vol


def _wmo_components(p, wa):
    model = int(p["imodele_wmo"])
    c = [0.] * 4
    if model == 1:
        c[0], c[1], c[3] = _F(0.70), _F(0.29), _F(0.01)
    elif model == 2:
        c[1], c[2] = _F(0.05), _F(0.95)
    elif model == 3:
        c[0], c[1], c[3] = _F(0.17), _F(0.61), _F(0.22)
    elif model == 4:
        c = [float(p["c_wmo_dl"]), float(p["c_wmo_ws"]), float(p["c_wmo_oc"]), float(p["c_wmo_so"])]
    else:
        raise AerosolError("-AER.WMO.Model must be 1..4")
    r, v2, mr, mi, vol = init_param_wmo(wa)
    n = [c[i] / vol[i] for i in range(4)]
    ntot = 0.
    for x in n:
        ntot = ntot + x
    alphaf = (4000., 50., 800., 10.)                        # CTE_ALPHAMAX_WMO_DL/WS/OC/SO, SOS.h:122-125
    # the reference skips a component on C(I) = 0 (:1398)
    return [((n[i] / ntot) if c[i] != 0. else 0., mr[i], mi[i], r[i], v2[i], alphaf[i]) for i in range(4)]


def init_param_sf(wa, rh):
    """SOS_INIT_PARAMSF (SOS_AEROSOLS.F:3557-3842): modal radii (interpolated in relative humidity), ln-variances and
    refractive indices (interpolated in wavelength, then humidity) of the five Shettle & Fenn components
    small rural, large rural, small urban, large urban, oceanic."""
    rows = _rows(_fic("Data_SF_cor_2015_12_16"))
    v2 = [x * math.log(10.) for x in rows[0][:5]]
    r = [0.] * 5
    tab = rows[1:]
    rh1, rm1 = tab[0][0], tab[0][1:6]
    rh2 = None
    cpt = 1
    if rh1 == rh:
        r = list(rm1)
    else:
        for row in tab[1:]:
            rh2, rm2 = row[0], row[1:6]
            cpt += 1
            if rh1 < rh <= rh2:
                r = [_interpol(rm1[i], rm2[i], rh1, rh2, rh) for i in range(5)]
                break
            rh1, rm1 = rh2, rm2
        else:
            raise AerosolError("Error while reading the Shettle&Fenn component datafile (relative humidity outside the table)")
    files = ("IRefrac_SR_cor_2015_12_16", "IRefrac_LR", "IRefrac_SU_cor_2015_12_16", "IRefrac_LU_cor_2015_12_16",
             "IRefrac_OM_cor_2015_12_16")
    mr, mi = [0.] * 5, [0.] * 5
    for i, name in enumerate(files):
        tabi = _rows(_fic(name))
        found = False
        for lo, hi in zip(tabi[:-1], tabi[1:]):
            wa1, wa2 = lo[0], hi[0]
            if wa1 <= wa <= wa2:
                col = lambda row, h: (row[1 + 2 * (h - 1)], row[2 + 2 * (h - 1)])      # MR(h), MI(h), h = 1..8
                if cpt == 1:
                    a = _interpol(col(lo, cpt)[0], col(hi, cpt)[0], wa1, wa2, wa)
                    b = _interpol(col(lo, cpt)[1], col(hi, cpt)[1], wa1, wa2, wa)
                else:
                    a1 = _interpol(col(lo, cpt - 1)[0], col(hi, cpt - 1)[0], wa1, wa2, wa)
                    a2 = _interpol(col(lo, cpt)[0], col(hi, cpt)[0], wa1, wa2, wa)
                    a = _interpol(a1, a2, rh1, rh2, rh)
                    b1 = _interpol(col(lo, cpt - 1)[1], col(hi, cpt - 1)[1], wa1, wa2, wa)
                    b2 = _interpol(col(lo, cpt)[1], col(hi, cpt)[1], wa1, wa2, wa)
                    b = _interpol(b1, b2, rh1, rh2, rh)
                mr[i], mi[i] = _round_mr_mi(a, b)
                found = True
                break
        if not found:
            break                                            # END=999: the remaining components keep zero indices
    return r, v2, mr, mi


def _sf_components(p, wa):
    model = int(p["imodele_sf"])
    ni = [0.] * 5
    if model == 1:
        ni[0] = 1.0
    elif model == 2:
        ni[2], ni[3] = _F(0.999875), _F(0.000125)
    elif model == 3:
        ni[0], ni[4] = _F(0.99), _F(0.01)
    elif model == 4:
        ni[0], ni[4] = _F(0.995), _F(0.005)
    else:
        raise AerosolError("-AER.SF.Model must be 1..4")
    r, v2, mr, mi = init_param_sf(wa, float(p["rh"]))
    out = []
    for i in range(5):
        if i == 0:
            af = 70.                                        # CTE_ALPHAMAX_SF_SR
        elif i == 2:
            af = 90.                                        # CTE_ALPHAMAX_SF_SU
        else:
            af = float(np.float32(100 + 100 * math.trunc(2. * math.pi * _rmax_lnd(r[i], v2[i]) / (100. * wa)))) if ni[i] else 0.
        out.append((ni[i], mr[i], mi[i], r[i], v2[i], af))
    return out


def _after_colon(line):
    return line[line.index(":") + 1:].split()


def _external_phase_functions(path, xmu):
    """IMOD = 4 (SOS_AEROSOLS.F:2143-2280): extinction and scattering cross sections and F11, -F12/F11, F22/F11, F33/F11
    versus scattering angle from the user's file; spline interpolation in cos(angle) to the Mie angle set."""
    from .absorption import _spline, _splint
    if not os.path.exists(path):
        raise AerosolError("-AER.ExtData file %s not found" % path)
    with open(path) as f:
        lines = f.read().splitlines()
    kmat1 = float(_after_colon(lines[0])[0].replace("D", "E").replace("d", "e"))
    kmat2 = float(_after_colon(lines[1])[0].replace("D", "E").replace("d", "e"))
    nang = int(_after_colon(lines[2])[0])
    if nang > 200:
        raise AerosolError("too many angles in the -AER.ExtData file (CTE_MAXNB_ANG_EXT = 200)")
    rows = []
    for ln in lines[4:4 + nang]:
        t = [float(x.replace("D", "E").replace("d", "e")) for x in ln.replace(",", " ").split()[:5]]
        if len(t) < 5:
            raise AerosolError("error while reading the -AER.ExtData file")
        rows.append(t)
    if len(rows) < nang:
        raise AerosolError("unexpected end of the -AER.ExtData file")
    a = np.array(rows)
    mu = np.cos(a[:, 0] * math.pi / 180.)
    f11 = a[:, 1]
    f12, f22, f33 = -a[:, 2] * f11, a[:, 3] * f11, a[:, 4] * f11
    # SOS_INTERPO_SPLINT sorts the nodes by an exchange sort; equal abscissae do not occur for distinct angles
    order = np.argsort(mu, kind="stable")
    x = mu[order]
    out = []
    for y in (f11, f12, f22, f33):
        ys = y[order]
        dy1 = (ys[1] - ys[0]) / (x[1] - x[0])
        dyn = (ys[-1] - ys[-2]) / (x[-1] - x[-2])
        d2 = _spline(x, ys, dy1, dyn)
        out.append(np.array([_splint(x, ys, d2, xv) for xv in xmu]))
    p11, p12, p22, p33 = out
    return kmat1, kmat2, p11, p12, p22, p33


def read_mixture_file(path):
    """The -AER.DefMixture file (SOS_AEROSOLS.F:2296-2372): number of modes, then per mode the size-distribution type and
    parameters, the refractive indices at the simulation and reference wavelengths and the share of the reference AOT."""
    if not os.path.exists(path):
        raise AerosolError("-AER.DefMixture file %s not found" % path)
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines()]
    it = iter(lines)
    val = lambda: _after_colon(next(it))[0]
    num = lambda: float(val().replace("D", "E").replace("d", "e"))
    modes = []
    for _ in range(int(val())):
        kind = val().strip("'\"")
        if kind == "LND":
            r = num(); v = num()
            m = dict(igranu=1, v1=r, v2=v, v3=-999.0)
        elif kind == "JUNGE":
            slope = num(); rmin = num(); rmax = num()
            m = dict(igranu=2, v1=rmin, v2=slope, v3=rmax)
        else:
            raise AerosolError("mixture file: the size distribution must be LND or JUNGE")
        m["rn_wa"], m["in_wa"], m["rn_waref"], m["in_waref"], m["rate"] = num(), num(), num(), num(), num()
        modes.append(m)
    return modes


def _user_mixture(p, wa, xmu, device):
    """IMOD = 5 (SOS_AEROSOLS.F:2289-2770): modes weighted so that each holds its share of the optical thickness at the
    reference wavelength."""
    modes = read_mixture_file(str(p["ficmixture_aer"]).strip())
    if len(modes) > 20:
        raise AerosolError("too many modes in the mixture file (CTE_MAX_NB_MODE_MIXTURE = 20)")
    waref, ta_ref = float(p["waref_aot"]), float(p["aot_ref"])
    aot = [ta_ref * m["rate"] for m in modes]
    som = 0.
    for m in modes:
        som = som + m["rate"]
    if abs(som - 1.) > 0.000001:                             # CTE_GAP_TOLER_SUM_RATES (double-precision context, SOS.h:184)
        raise AerosolError("mixture file: the sum of the AOT rates is not equal to 1")
    if som != 1.:
        aot = [x / som for x in aot]
    def alphaf_of(m):
        rmax = _rmax_lnd(m["v1"], m["v2"]) if m["igranu"] == 1 else m["v3"]
        return _alphaf(rmax, WAMIN)
    coef, tot = [], 0.
    for m, tau in zip(modes, aot):
        k1 = size_integral(xmu, m["rn_waref"], m["in_waref"], alphaf_of(m), m["igranu"], m["v1"], m["v2"], m["v3"], waref, device)[0]
        coef.append(tau / k1)
        tot = tot + coef[-1]
    coef = [c / tot for c in coef]
    w = len(xmu)
    kmat1 = kmat2 = 0.
    p11, p12, p33 = np.zeros(w), np.zeros(w), np.zeros(w)
    for m, c in zip(modes, coef):
        rn, in_ = (m["rn_waref"], m["in_waref"]) if wa == waref else (m["rn_wa"], m["in_wa"])
        k1, k2, _, a11, a12, a33 = size_integral(xmu, rn, in_, alphaf_of(m), m["igranu"], m["v1"], m["v2"], m["v3"], wa, device)
        kmat1 = kmat1 + c * k1
        kmat2 = kmat2 + c * k2
        p11 = p11 + a11 * c * k2
        p12 = p12 + a12 * c * k2
        p33 = p33 + a33 * c * k2
    return kmat1, kmat2, p11 / kmat2, p12 / kmat2, p33 / kmat2


def aerosols(p, wa, ta, nb_gauss_mie, os_nb, *, at_waref=False, device=0):
    """SOS_AEROSOLS for the wavelength wa.  p: the sos_proc keyword dictionary (run_sos.SOS_PROC_KWARGS names); at_waref
    selects the refractive indices of the reference wavelength (the first of the two calls SOS_PROC makes when
    WA_SIMU != WAREF, SOS_PROC.F:2893 / :3028).  Returns the content of Aerosols.txt:
    dict(alpha, beta, gamma, zeta [os_nb+1], a_tronc, piz, piztr (as printed: F9.5), kmat1, kmat2, coef_tronca (full precision))."""
    imod = int(p["imod_aer"])
    itronc = int(p["itronc_aer"])
    xmu, xhr = mie_angles(nb_gauss_mie)
    p22 = None
    sfx = "ref" if at_waref else ""
    if ta == 0.0:
        z = np.zeros(os_nb + 1)
        return dict(alpha=z, beta=z.copy(), gamma=z.copy(), zeta=z.copy(), a_tronc=0.0, piz=0.0, piztr=0.0, kmat1=0.0, kmat2=0.0,
                    coef_tronca=0.0)
    if imod == 0:
        rn, in_ = _round_index(p["rn_wa" + sfx], p["in_wa" + sfx])
        igranu = int(p["igranu"])
        if igranu == 1:
            v1, v2, v3 = p["lnd_radius_mmd_aer"], p["lnd_lnvar_mmd_aer"], -999.0
            rmax = _rmax_lnd(v1, v2)
        elif igranu == 2:
            v1, v2, v3 = p["jd_rmin_mmd_aer"], p["jd_slope_mmd_aer"], p["jd_rmax_mmd_aer"]
            rmax = v3
        else:
            raise AerosolError("-AER.MMD.SDtype must be 1 (LND) or 2 (Junge)")
        af = _alphaf(rmax, WAMIN)                     # mono-modal model: grid sized for the shortest wavelength (:1158)
        kmat1, kmat2, _, p11, p12, p33 = size_integral(xmu, rn, in_, af, igranu, v1, v2, v3, wa, device)
    elif imod == 3:
        modes = []
        for m in ("cm", "fm"):
            rn, in_ = _round_index(p["bmd_%s_mrwa%s" % (m, sfx)], p["bmd_%s_miwa%s" % (m, sfx)])
            modes.append(dict(rn=rn, in_=in_, r=p["bmd_%s_rmodal" % m], v=p["bmd_%s_var" % m]))
        vcdef = int(p["mode_param_bilnd"])
        if vcdef == 1:
            cvi = [p["user_cv_coarse"], p["user_cv_fine"]]
        elif vcdef == 2:
            # volume concentrations from the coarse-mode share of the optical thickness at the REFERENCE wavelength
            waref, kref = p["waref_aot"], []
            for m, md in zip(("cm", "fm"), modes):
                rnr, inr = _round_index(p["bmd_%s_mrwaref" % m], p["bmd_%s_miwaref" % m])
                kref.append(size_integral(xmu, rnr, inr, _alphaf(_rmax_lnd(md["r"], md["v"]), waref), 1, md["r"], md["v"], -999.0,
                                          waref, device)[0])
            rt, ta_ref = p["rtauct_waref"], p["aot_ref"]
            cvi = [(rt * ta_ref) / kref[0], ((1. - rt) * ta_ref) / kref[1]]
        else:
            raise AerosolError("-AER.BMD.VCdef must be 1 or 2")
        ntot = cvi[0] + cvi[1]
        cvi = [cvi[0] / ntot, cvi[1] / ntot]
        kmat1 = kmat2 = 0.
        w = len(xmu)
        p11, p12, p33 = np.zeros(w), np.zeros(w), np.zeros(w)
        for c, md in zip(cvi, modes):
            if c == 0.:
                continue
            k1, k2, _, a11, a12, a33 = size_integral(xmu, md["rn"], md["in_"], _alphaf(_rmax_lnd(md["r"], md["v"]), wa), 1, md["r"],
                                                     md["v"], -999.0, wa, device)
            kmat1 = kmat1 + c * k1
            kmat2 = kmat2 + c * k2
            p11 = p11 + c * a11 * k2
            p12 = p12 + c * a12 * k2
            p33 = p33 + c * a33 * k2
        p11, p12, p33 = p11 / kmat2, p12 / kmat2, p33 / kmat2
    elif imod == 1:
        kmat1, kmat2, p11, p12, p33 = _mix(_wmo_components(p, wa), xmu, wa, device)
    elif imod == 2:
        kmat1, kmat2, p11, p12, p33 = _mix(_sf_components(p, wa), xmu, wa, device)
    elif imod == 4:
        kmat1, kmat2, p11, p12, p22, p33 = _external_phase_functions(str(p["ficextdata_aer"]).strip(), xmu)
    elif imod == 5:
        kmat1, kmat2, p11, p12, p33 = _user_mixture(p, wa, xmu, device)
    else:
        raise AerosolError("-AER.Model must be 0..5")
    if getattr(_TLS, "collect", None) is not None:      # COLLECT mode: the size integrals asked for are recorded, nothing more
        return None
    if p22 is None:
        p22 = p11.copy()                                # spherical particles (:1234, :1495, :1688, :2118, :2762)
    if getattr(_TLS, "defer_expansion", False):         # aerosols_many: the expansions of a chunk of wavelengths are formed together
        return ("expansion", itronc, xmu, xhr, os_nb, p11, p12, p22, p33, kmat1, kmat2)
    return _aerosols_file(decompo_legendre(itronc, xmu, xhr, os_nb, p11, p12, p22, p33), kmat1, kmat2)


def _aerosols_file(d, kmat1, kmat2):
    """The content of Aerosols.txt from the expansion d and the cross sections (the tail of SOS_AEROSOLS)."""
    piz = kmat2 / kmat1
    ct = d["coef_tronca"]
    piztr = piz * (1. - ct / 2.) / (1. - piz * ct / 2.)
    # what SOS_PREPA_OS reads back from Aerosols.txt: E15.8 coefficients, F9.5 truncation coefficient and albedo
    q = _round_sig(np.stack([d["alpha"], d["beta"], d["gamma"], d["zeta"]]), 8)      # (element-wise: one call for the four)
    a_f, piztr_f = float("%9.5f" % ct), float("%9.5f" % piztr)
    return dict(alpha=q[0], beta=q[1], gamma=q[2], zeta=q[3], a_tronc=a_f, piztr=piztr_f,
                piz=piztr_f / (1 + 0.5 * a_f * (piztr_f - 1)), kmat1=kmat1, kmat2=kmat2, coef_tronca=ct)


def aerosols_many(calls, nb_gauss_mie, os_nb, device=0):
    """aerosols(p, wa, ta, ...) for a list of (p, wa, ta) -- the wavelengths of a chunk of a spectrum -- with the Legendre
    expansions of all of them formed in one vectorised pass (decompo_legendre_many).  Returns a list parallel to `calls`; an
    entry is None when its call raised (the caller repeats it on its own and reports).  Outputs equal aerosols()'s bit for bit."""
    out = [None] * len(calls)
    todo = collections.OrderedDict()
    _TLS.defer_expansion = True
    try:
        for i, (p, wa, ta) in enumerate(calls):
            try:
                r = aerosols(p, wa, ta, nb_gauss_mie, os_nb, at_waref=False, device=device)
            except Exception:
                continue
            if isinstance(r, tuple):
                _, itronc, xmu, xhr, nb, p11, p12, p22, p33, k1, k2 = r
                todo.setdefault((itronc, xmu.tobytes(), xhr.tobytes(), nb), []).append((i, xmu, xhr, p11, p12, p22, p33, k1, k2))
            else:
                out[i] = r
    finally:
        _TLS.defer_expansion = False
    for (itronc, _, _, nb), members in todo.items():
        try:
            ds = decompo_legendre_many(itronc, members[0][1], members[0][2], nb, np.stack([m[3] for m in members]),
                                       np.stack([m[4] for m in members]), np.stack([m[5] for m in members]),
                                       np.stack([m[6] for m in members]))
        except Exception:
            continue
        for m, d in zip(members, ds):
            try:
                out[m[0]] = _aerosols_file(d, m[7], m[8])
            except Exception:
                pass
    return out
