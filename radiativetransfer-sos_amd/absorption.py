"""Gas absorption of one wavelength by the CKD method: the steps of the reference that turn a wavelength and an
atmosphere type into the per-bin absorption optical-depth profiles SOS_PROFILE takes.

    prepa_absprofile  <- SOS_PREPA_ABSPROFILE  src/SOS_PREPA_ABSPROFILE.F:248   gas amounts per layer, CKD tables, LAMB1
    datatm            <- DATATM                src/SOS_SUB_TRS.F:908            predefined / user atmosphere -> mixing ratios
    read_ckd_coeff    <- READ_CKD_COEFF        src/SOS_SUB_TRS.F:481            one gas, the 50-interval file holding NU
    coeff_abs_ckd     <- COEFF_ABS_CKD         src/SOS_SUB_TRS.F:171            k_i(P, T[, c_H2O]) of one layer
    layer_tables      <- the K / J loops of SOS_ABSPROFILE src/SOS_ABSPROFILE.F:325-353, hoisted: XK(gas, term, layer) is the
                         same for every bin, only the choice of the term per gas differs
    bins              <- the eight nested loops of SOS_PROC.F:3381-3404, 3459-3487 (ckd.ckd_bin_weights)

The per-bin part (sum over the gases, cumulative transmission, -ln: SOS_ABSPROFILE.F:355-371) runs on the device for all
bins at once (csrc/profile.hip k_absprofile, C ABI sosgpu_absprofile); `absprofile_host` is the same loop in numpy for one
bin, used by the single-profile mode -SOS.AbsModeCKD 2 bookkeeping and by tests.

Data: CKD coefficient files and SO2-NO2 are read from $SOS_ABS_ROOT/fic like the reference does (GETENV, SOS_SUB_TRS.F:616);
the six predefined atmospheres come from data/afgl_atmospheres.npz (scripts/make_afgl_tables.py).  REAL*4 literals of the
Fortran are widened exactly as the compiler does (`_F`)."""
import functools
import os
import threading

import numpy as np

from .ckd import GASES, ckd_bin_weights

_F = lambda x: float(np.float32(x))
NBABS, NLEVEL, NBCOL = 8, 50, 13                              # SOS.h:246,250,254
CKD_NAI_MAX, CKD_NB_NU_PER_FILE, CKD_NUMAX, CKD_NUMIN = 5, 50, 27500, 2500      # SOS.h:279,292,290,291
TAUABS_MAX = _F(999.)                                         # SOS.h:297
ATMOCM = np.array([_F(3.410E+22), _F(1.395E+22), _F(1.279E+22), _F(1.395E+22), _F(2.192E+22), _F(3.837E+22), _F(1.918E+22),
                   _F(1.3340E+22)])                           # SOS_PREPA_ABSPROFILE.F:371-373
PDSMOL = np.array([18., 44., 48., 44., 28., 16., 32., 46.])
HERE = os.path.dirname(os.path.abspath(__file__))


class AbsorptionError(RuntimeError):
    """The reference's IER = -1 of SOS_PREPA_ABSPROFILE / READ_CKD_COEFF / COEFF_ABS_CKD, with its message."""


def fic_root():
    root = os.environ.get("SOS_ABS_ROOT", "")
    if not root:
        raise AbsorptionError("SOS_ABS_ROOT is not defined (SOS_PREPA_ABSPROFILE ERROR_925)")
    return os.path.join(root, "fic")


def predefined_atmosphere(absprofil):
    """DONUSER columns 1..11 of atmosphere type 1..6 as DATATM fills them (SOS_SUB_TRS.F:944-975, PSURF scaling apart)."""
    if not 1 <= absprofil <= 6:
        raise AbsorptionError("-AP.AbsProfile.Type must be 0..7")
    with np.load(os.path.join(HERE, "data", "afgl_atmospheres.npz")) as z:
        return z["donuser"][absprofil - 1].astype(np.float64)


def read_user_profile(path):
    """-AP.AbsProfile.UserFile (ABSPROFIL = 0, SOS_PREPA_ABSPROFILE.F:441-447): 50 rows `index, 13 columns`."""
    rows = []
    with open(path) as f:
        for line in f:
            v = line.replace("D", "E").split()
            if len(v) >= 1 + NBCOL:
                rows.append([float(x) for x in v[1:1 + NBCOL]])
            if len(rows) == NLEVEL:
                break
    if len(rows) != NLEVEL:
        raise AbsorptionError("ERROR while reading user profile by SOS_PREPAPROFIL")
    return np.array(rows)


def datatm(absprofil, userprofil, psurf):
    """DATATM: fills / rescales the profile table (pressure scaled to PSURF when it is given) and converts the volume mixing
    ratios (ppmv) to mass mixing ratios RO[8][50] per LEVEL (NO2 is added by the caller).  Returns (ro, p, t, alt)."""
    u = userprofil
    if absprofil > 0:
        u[:, :11] = predefined_atmosphere(absprofil)
    coef = psurf / u[0, 1] if psurf > 0. else 1.0
    alt, t = u[:, 0].copy(), u[:, 2].copy()
    if absprofil > 0:
        p = u[:, 1].copy()          # predefined atmospheres: only the TABLE is rescaled (SOS_SUB_TRS.F:961-966), P(J) --
        u[:, 1] = p * coef          # used for the layer amounts DP(J) -- keeps the model's own pressures
    else:
        p = u[:, 1] * coef          # user profile: P(J) is rescaled, the table is not (SOS_SUB_TRS.F:931-938)
    e6, m = _F(1.0E-06), _F(28.97)
    ro = np.zeros((NBABS, NLEVEL))
    ro[1] = u[:, 4] * e6 * 44.0 / m
    ro[3] = u[:, 6] * e6 * 44.0 / m
    ro[4] = u[:, 7] * e6 * 28.0 / m
    ro[5] = u[:, 8] * e6 * 16.0 / m
    ro[6] = u[:, 9] * e6 * 32.0 / m
    h2o = u[:, 3] * e6 * 18.0 / m
    ro[0] = h2o / (1 + h2o)
    ro[2] = u[:, 5] * e6 * 48.0 / m
    return ro, p, t, alt


def ckd_file_name(nabs, nu, nustep):
    """coef_<GAS>_<numax>_<numin>_<step>cmm1 of the 50-interval file holding NU (SOS_SUB_TRS.F:616-698)."""
    numin = int(CKD_NUMAX - CKD_NB_NU_PER_FILE * nustep)
    while numin > nu:
        numin = int(numin - CKD_NB_NU_PER_FILE * nustep)
    numax = int(numin + CKD_NB_NU_PER_FILE * nustep)
    step = int(nustep)
    return os.path.join("COEFF_CKD", "%dcmm1" % step, "coef_%s_%d_%d_%dcmm1" % (GASES[nabs - 1], numax, numin, step)), numax, numin


def read_ckd_coeff(nabs, nu, nustep, root=None):
    """READ_CKD_COEFF for gas nabs (1-based): dict(numax, numin, tab_temp, tab_pres, tab_conc (H2O), nexp[nwa],
    ai[nwa][5], ki[nwa][5][NP][NT] (H2O: [nwa][5][NC][NP][NT])).  List-directed reads: one record per READ.
    A file holds 50 spectral intervals (the H2O ones take 0.25 s to parse): parsed files are kept (the last 512 -- a whole
    spectrum of the 10 cm-1 tables is 400 files, about 0.5 GB parsed --, keyed by path, size and modification time), so a
    hyperspectral loop re-reads nothing."""
    if nustep not in (1, 5, 10):
        raise AbsorptionError("The required spectral resolution is not supported : %g cm-1" % nustep)
    rel, numax_f, numin_f = ckd_file_name(nabs, nu, nustep)
    path = os.path.join(root or fic_root(), rel)
    try:
        st = os.stat(path)
    except OSError:
        raise AbsorptionError("Error while opening the file of CKD coefficients\nFile :%s" % path)
    return _read_ckd_file(path, nabs, nustep, numax_f, numin_f, st.st_size, st.st_mtime_ns)


@functools.lru_cache(maxsize=512)
def _read_ckd_file(path, nabs, nustep, numax_f, numin_f, _size, _mtime):
    try:
        with open(path) as f:
            lines = f.read().splitlines()
    except OSError:
        raise AbsorptionError("Error while opening the file of CKD coefficients\nFile :%s" % path)
    pos = 21 if nabs == 1 else 18

    def rec(n=None):
        nonlocal pos
        vals = []
        while n is None or len(vals) < n:
            vals += lines[pos].replace(",", " ").split()
            pos += 1
            if n is None:
                break
        return vals

    numax, numin, res = (float(v) for v in rec(3)[:3])
    if res != nustep or numax != numax_f or numin != numin_f:
        raise AbsorptionError("Not consistent spectral range / resolution in file :%s" % path)
    nwa = int((numax - numin) / res)
    nt = int(rec(1)[0]); tab_t = np.array([float(v) for v in rec(nt)[:nt]])
    npr = int(rec(1)[0]); tab_p = np.array([float(v) for v in rec(npr)[:npr]])
    nc, tab_c = 1, None
    if nabs == 1:
        nc = int(rec(1)[0]); tab_c = np.array([float(v) for v in rec(nc)[:nc]])
    nexp = np.ones(nwa, dtype=np.int32)
    ai = np.zeros((nwa, CKD_NAI_MAX))
    ki = np.zeros((nwa, CKD_NAI_MAX, nc, npr, nt))
    lead = 3 if nabs == 1 else 2
    for iwa in range(nwa):
        nmax = int(rec(6)[5])
        if nmax == 0:
            ai[iwa, 0] = 1.0
            continue
        nexp[iwa] = nmax
        ai[iwa, :nmax] = [float(v) for v in rec(nmax)[:nmax]]
        nrow = nmax * nc * npr
        # (one conversion for the whole block: rows of `lead` indices + nt coefficients; a row of another shape -- never seen
        #  in the reference's tables -- falls back to the per-row form)
        flat = np.array(" ".join(lines[pos:pos + nrow]).split(), dtype=np.float64)
        if flat.size == nrow * (lead + nt):
            blk = flat.reshape(nrow, lead + nt)[:, lead:]
        else:
            blk = np.array([[float(v) for v in lines[pos + r].split()[lead:lead + nt]] for r in range(nrow)])
        pos += nrow
        ki[iwa, :nmax] = blk.reshape(nmax, nc, npr, nt)
    if nabs != 1:
        ki = ki[:, :, 0]
    return dict(numax=numax, numin=numin, tab_temp=tab_t, tab_pres=tab_p, tab_conc=tab_c, nexp=nexp, ai=ai, ki=ki)


def _interpol(y1, y2, x1, x2, x):
    return ((y2 - y1) / (x2 - x1)) * (x - x2) + y2                       # SOS_INTERPOL, SOS_AEROSOLS.F:3862


def _spline(x, y, dy1, dyn):
    """SOS_SPLINE (SOS_AEROSOLS.F:4976-5010): second derivatives of the natural-cubic-style spline with end slopes."""
    n = len(x)
    d2, u = np.zeros(n), np.zeros(n)
    if dy1 > _F(.99E30):
        d2[0] = 0.; u[0] = 0.
    else:
        d2[0] = -0.5
        u[0] = (3. / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - dy1)
    for k in range(1, n - 1):
        sig = (x[k] - x[k - 1]) / (x[k + 1] - x[k - 1])
        p = sig * d2[k - 1] + 2.
        d2[k] = (sig - 1.) / p
        u[k] = (6. * ((y[k + 1] - y[k]) / (x[k + 1] - x[k]) - (y[k] - y[k - 1]) / (x[k] - x[k - 1])) / (x[k + 1] - x[k - 1])
                - sig * u[k - 1]) / p
    if dyn > _F(.99E30):
        qn = 0.; un = 0.
    else:
        qn = 0.5
        un = (3. / (x[n - 1] - x[n - 2])) * (dyn - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]))
    d2[n - 1] = (un - qn * u[n - 2]) / (qn * d2[n - 2] + 1.)
    for k in range(n - 2, -1, -1):
        d2[k] = d2[k] * d2[k + 1] + u[k]
    return d2


def _splint(x, y, d2, xv):
    """SOS_SPLINT (SOS_AEROSOLS.F:5060-5079)."""
    klo, khi = 0, len(x) - 1
    while khi - klo > 1:
        k = (khi + klo + 2) // 2 - 1                      # (KHI+KLO)/2 on 1-based indices
        if x[k] > xv:
            khi = k
        else:
            klo = k
    h = x[khi] - x[klo]
    if h == 0.:
        raise AbsorptionError("ERROR for SPLINT interpolation")
    a = (x[khi] - xv) / h
    b = (xv - x[klo]) / h
    # A**3, H**2 with integer exponents are products in the compiled Fortran
    return a * y[klo] + b * y[khi] + ((a * a * a - a) * d2[klo] + (b * b * b - b) * d2[khi]) * (h * h) / 6.


def interpo_splint(xin, yin, xv):
    """SOS_INTERPO_SPLINT for one abscissa (the tables are already ascending)."""
    dy1 = (yin[1] - yin[0]) / (xin[1] - xin[0])
    dyn = (yin[-1] - yin[-2]) / (xin[-1] - xin[-2])
    return _splint(xin, yin, _spline(xin, yin, dy1, dyn), xv)


def coeff_abs_ckd(nabs, ki, tab_pres, tab_temp, tab_conc, prs, tmp, conc):
    """COEFF_ABS_CKD for one exponential term: ki [NP][NT] (H2O: [NC][NP][NT]).  Returns (xk, prs, tmp, conc) -- the
    reference clamps PRS / TMP / CONC_H2O IN PLACE, and SOS_ABSPROFILE carries the clamped values to the next gas."""
    tmp = min(max(tmp, tab_temp[0]), tab_temp[-1])
    if prs <= tab_pres[0]:
        return 0.0, prs, tmp, conc
    prs = min(prs, tab_pres[-1])
    if tab_conc is not None:
        conc = min(max(conc, tab_conc[0]), tab_conc[-1])
    npr = len(tab_pres)
    ip = 0
    while tab_pres[ip] <= prs and ip < npr - 1:
        ip += 1
    ip -= 1
    if nabs == 1:
        nc = len(tab_conc)
        ic = 0
        while tab_conc[ic] <= conc and ic < nc - 1:
            ic += 1
        ic -= 1
        xkh = _interpol(ki[ic], ki[ic + 1], tab_conc[ic], tab_conc[ic + 1], conc)            # [NP][NT]
        xki = _interpol(xkh[ip], xkh[ip + 1], tab_pres[ip], tab_pres[ip + 1], prs)
    else:
        xki = _interpol(ki[ip], ki[ip + 1], tab_pres[ip], tab_pres[ip + 1], prs)            # [NT]
    xk = interpo_splint(tab_temp, xki, tmp)
    if xk < 0.:
        nt = len(tab_temp)
        it = 0
        while tab_temp[it] <= tmp and it < nt - 1:
            it += 1
        it -= 1
        xk = _interpol(xki[it], xki[it + 1], tab_temp[it], tab_temp[it + 1], tmp)
        if xk < 0.:
            raise AbsorptionError("COEFF_ABS_CKD : ERROR_923 : Calculations give ki < 0 : uncorrect value!")
    return float(xk), prs, tmp, conc


@functools.lru_cache(maxsize=8)
def _atmosphere(psurf, h2o, o3, co2, ch4, absprofil, ficabsprofil, root, _stamp):
    """The wavelength-independent half of SOS_PREPA_ABSPROFILE (SOS_PREPA_ABSPROFILE.F:441-553): profile table, layer amounts
    of the eight gases, scaling to the user's H2O / O3 / CO2 / CH4 amounts.  Cached: a spectrum of calls shares it (the arrays
    are read-only from here on).  _stamp: (size, mtime) of the user profile file, so that an edited file is re-read."""
    user = np.zeros((NLEVEL, NBCOL))
    if absprofil == 0:
        user = read_user_profile(ficabsprofil)
    else:
        try:
            with open(os.path.join(root, "SO2-NO2")) as f:
                rows = [ln.split() for ln in f.read().splitlines()[:NLEVEL]]
            user[:, 12] = [float(r[0]) for r in rows]
            user[:, 11] = [float(r[1]) for r in rows]
        except (OSError, IndexError, ValueError):
            raise AbsorptionError("SOS_PREPA_ABSPROFILE : ERROR_927: Error while reading the file fic/SO2-NO2")
    ro, p, t, altc = datatm(absprofil, user, psurf)
    ro[7] = user[:, 11] * _F(1.0E-06) * 46 / _F(28.9)
    co2_def = ro[1, 0] * _F(28.97) / _F(44.0E-06)
    ch4_def = ro[5, 0] * _F(28.97) / _F(16.0E-06)
    altabs = altc[::-1].copy()
    dp = p[:-1] - p[1:]
    ro[:, :-1] = dp[None, :] * (ro[:, :-1] + ro[:, 1:]) / 2.0 * ATMOCM[:, None]

    def serial_sum(x):
        s = 0.
        for v in x:
            s = s + float(v)
        return s

    if h2o >= 0.:
        q = serial_sum(ro[0]) / _F(6.022E+23) * PDSMOL[0]
        ro[0] = ro[0] * h2o / q
        user[:, 3] = user[:, 3] * h2o / q
    if o3 >= 0.:
        o3 = o3 / 1000.
        q = serial_sum(ro[2]) / _F(6.022E+23) * PDSMOL[2]
        q = q * _F(466.23)
        ro[2] = ro[2] * o3 / q
        user[:, 5] = user[:, 5] * o3 / q
    if co2 >= 0.:
        ro[1] = ro[1] * co2 / co2_def
        user[:, 4] = user[:, 4] * co2 / co2_def
    if ch4 >= 0.:
        ro[5] = ro[5] * ch4 / ch4_def
        user[:, 8] = user[:, 8] * ch4 / ch4_def
    for a in (user, ro, altabs):
        a.setflags(write=False)
    return user, ro, altabs


# ---- gas tables of many wavelengths prepared together (run_sos.sos_spectrum) -----------------------------------------------
# prefetch_gas_tables runs SOS_PREPA_ABSPROFILE for the wavelengths of a chunk and interpolates the layer tables of ALL of them
# in one vectorised pass (layer_tables_many); prepa_absprofile / layer_tables calls of this thread with the same arguments
# then return the prepared objects.  Per thread; dropped by the caller at the end of the chunk.
_PREFETCH = threading.local()


def prefetch_gas_tables(requests):
    """requests: argument tuples of prepa_absprofile (wa, nustep, psurf, h2o, o3, co2, ch4, absprofil, ficabsprofil).  A
    request that fails is left to the call that owns it."""
    preps = {}
    root = None
    try:
        root = fic_root()
    except AbsorptionError:
        return 0
    for r in requests:
        key = tuple(r) + (root,)
        if key in preps:
            continue
        try:
            preps[key] = prepa_absprofile(*r, root=root)
        except (AbsorptionError, OSError, ValueError):
            pass
    try:
        layer_tables_many(list(preps.values()))
    except AbsorptionError:
        pass                                           # (raised again, by its owner, in layer_tables)
    _PREFETCH.preps = preps
    return len(preps)


def drop_prefetched_gas_tables():
    _PREFETCH.preps = None


def prepa_absprofile(wa, nustep, psurf, h2o, o3, co2, ch4, absprofil, ficabsprofil=None, root=None):
    """SOS_PREPA_ABSPROFILE.  Returns dict(nu, lamb1, altabs[50] descending, userprofil[50][13], ro[8][50] molecules/cm2
    per layer (J = 1 lowest layer; entry 50 keeps the level value, as in the reference), gas tables of the interval
    LAMB1: nexp[8], kdis_ai[5][8], ki[8] (per gas [5][NP][NT] / H2O [5][NC][NP][NT]), tab_pres, tab_temp, tab_conc)."""
    root = root or fic_root()
    pre = getattr(_PREFETCH, "preps", None)
    if pre:
        hit = pre.get((wa, nustep, psurf, h2o, o3, co2, ch4, absprofil, ficabsprofil, root))
        if hit is not None:
            return hit
    stamp = None
    if absprofil == 0:
        try:
            st = os.stat(ficabsprofil)
            stamp = (st.st_size, st.st_mtime_ns)
        except OSError:
            stamp = None
    else:
        try:
            st = os.stat(os.path.join(root, "SO2-NO2"))
            stamp = (st.st_size, st.st_mtime_ns)
        except OSError:
            stamp = None
    user, ro, altabs = _atmosphere(float(psurf), float(h2o), float(o3), float(co2), float(ch4), int(absprofil),
                                   ficabsprofil if absprofil == 0 else None, root, stamp)
    nu = _F(1.0E+4) / wa
    if nu > CKD_NUMAX or nu < CKD_NUMIN:
        raise AbsorptionError("The simulation wavelength is not included in the spectral range of CKD data")
    gas = [read_ckd_coeff(k, nu, nustep, root) for k in range(1, NBABS + 1)]
    last = gas[-1]
    lamb1 = 1 + int((last["numax"] - nu) / nustep)
    iw = lamb1 - 1
    return dict(nu=nu, lamb1=lamb1, altabs=altabs, userprofil=user, ro=ro, absprofil=absprofil,
                nexp=np.array([g["nexp"][iw] for g in gas], dtype=np.int32),
                kdis_ai=np.stack([g["ai"][iw] for g in gas], axis=1),            # [5][8]
                ki=[g["ki"][iw] for g in gas], tab_pres=last["tab_pres"], tab_temp=last["tab_temp"],
                tab_conc=gas[0]["tab_conc"])


def _spline_rows(x, y, dy1, dyn):
    """_spline for many tables at once: y, result [n_rows][n]; x [n] common abscissae; the end slopes per row.  Same statements
    in the same order, element by element (IEEE double arithmetic: identical to the scalar routine)."""
    n = len(x)
    d2 = np.zeros_like(y)
    u = np.zeros_like(y)
    big = dy1 > _F(.99E30)
    d2[:, 0] = np.where(big, 0., -0.5)
    u[:, 0] = np.where(big, 0., (3. / (x[1] - x[0])) * ((y[:, 1] - y[:, 0]) / (x[1] - x[0]) - dy1))
    for k in range(1, n - 1):
        sig = (x[k] - x[k - 1]) / (x[k + 1] - x[k - 1])
        p = sig * d2[:, k - 1] + 2.
        d2[:, k] = (sig - 1.) / p
        u[:, k] = (6. * ((y[:, k + 1] - y[:, k]) / (x[k + 1] - x[k]) - (y[:, k] - y[:, k - 1]) / (x[k] - x[k - 1])) / (x[k + 1] - x[k - 1])
                   - sig * u[:, k - 1]) / p
    bign = dyn > _F(.99E30)
    qn = np.where(bign, 0., 0.5)
    un = np.where(bign, 0., (3. / (x[n - 1] - x[n - 2])) * (dyn - (y[:, n - 1] - y[:, n - 2]) / (x[n - 1] - x[n - 2])))
    d2[:, n - 1] = (un - qn * u[:, n - 2]) / (qn * d2[:, n - 2] + 1.)
    for k in range(n - 2, -1, -1):
        d2[:, k] = d2[:, k] * d2[:, k + 1] + u[:, k]
    return d2


def _bracket(tab, v):
    """Index i of the scalar search `while tab[i] <= v and i < n-1: i += 1; i -= 1` for every element of v."""
    i = np.searchsorted(tab, v, side="right")            # first index with tab[i] > v
    i = np.minimum(i, len(tab) - 1)
    return i - 1


def coeff_abs_ckd_rows(nabs, ki, tab_pres, tab_temp, tab_conc, prs, tmp, conc):
    """coeff_abs_ckd for arrays of layers (prs, tmp, conc [L]): returns (xk[L], prs, tmp, conc) with the clamped values, element
    by element what the scalar routine returns (tests/test_absorption.py compares the two)."""
    prs, tmp, conc = prs.copy(), tmp.copy(), conc.copy()
    tmp = np.minimum(np.maximum(tmp, tab_temp[0]), tab_temp[-1])
    low = prs <= tab_pres[0]                                  # no absorption above the table: state left as it is (but TMP)
    act = ~low
    xk = np.zeros(len(prs))
    if not act.any():
        return xk, prs, tmp, conc
    pa = np.minimum(prs[act], tab_pres[-1])
    ta = tmp[act]
    ca = conc[act]
    if tab_conc is not None:
        ca = np.minimum(np.maximum(ca, tab_conc[0]), tab_conc[-1])
    if not ki.any():
        # an all-zero table (NMAXAI = 0: the gas does not absorb in this interval -- most gases, most intervals): every
        # interpolation of zeros is zero; only the clamped state is carried on
        prs[act], conc[act] = pa, ca
        return xk, prs, tmp, conc
    ip = _bracket(tab_pres, pa)
    if nabs == 1:
        ic = _bracket(tab_conc, ca)
        k0, k1 = ki[ic], ki[ic + 1]                            # [La][NP][NT]
        c0, c1 = tab_conc[ic][:, None, None], tab_conc[ic + 1][:, None, None]
        xkh = ((k1 - k0) / (c1 - c0)) * (ca[:, None, None] - c1) + k1
        rows = np.arange(len(pa))
        a0, a1 = xkh[rows, ip], xkh[rows, ip + 1]
    else:
        a0, a1 = ki[ip], ki[ip + 1]                            # [La][NT]
    p0, p1 = tab_pres[ip][:, None], tab_pres[ip + 1][:, None]
    xki = ((a1 - a0) / (p1 - p0)) * (pa[:, None] - p1) + a1    # [La][NT]
    x = tab_temp
    dy1 = (xki[:, 1] - xki[:, 0]) / (x[1] - x[0])
    dyn = (xki[:, -1] - xki[:, -2]) / (x[-1] - x[-2])
    d2 = _spline_rows(x, xki, dy1, dyn)
    # _splint: bisection with (KHI+KLO)/2 on 1-based indices == the bracket of the ascending table for an in-range abscissa
    nt = len(x)
    klo = np.zeros(len(ta), dtype=np.int64); khi = np.full(len(ta), nt - 1, dtype=np.int64)
    while np.any(khi - klo > 1):
        k = (khi + klo + 2) // 2 - 1
        go = khi - klo > 1
        up = x[k] > ta
        khi = np.where(go & up, k, khi)
        klo = np.where(go & ~up, k, klo)
    rows = np.arange(len(ta))
    h = x[khi] - x[klo]
    if np.any(h == 0.):
        raise AbsorptionError("ERROR for SPLINT interpolation")
    aa = (x[khi] - ta) / h
    bb = (ta - x[klo]) / h
    val = aa * xki[rows, klo] + bb * xki[rows, khi] + ((aa * aa * aa - aa) * d2[rows, klo] + (bb * bb * bb - bb) * d2[rows, khi]) * (h * h) / 6.
    neg = val < 0.
    if neg.any():
        it = _bracket(x, ta)
        lin = ((xki[rows, it + 1] - xki[rows, it]) / (x[it + 1] - x[it])) * (ta - x[it + 1]) + xki[rows, it + 1]
        val = np.where(neg, lin, val)
        if np.any(val < 0.):
            raise AbsorptionError("COEFF_ABS_CKD : ERROR_923 : Calculations give ki < 0 : uncorrect value!")
    xk[act] = val
    prs[act], conc[act] = pa, ca
    return xk, prs, tmp, conc


def layer_tables_by_pair(prep):
    """layer_tables, one coeff_abs_ckd_rows call per (term, gas) pair with the clamped pressure / temperature / water-vapour
    state carried from gas to gas per layer as the reference carries it (the form of round 2; now the checker of layer_tables in
    tests/test_absorption.py, between it and the layer-by-layer layer_tables_scalar)."""
    u = prep["userprofil"]
    nl = NLEVEL
    xk = np.zeros((NBABS, CKD_NAI_MAX, nl - 1))
    j = np.arange(1, nl)
    lo, hi = nl - j - 1, nl - j                       # USERPROFIL(NLEVEL-J), USERPROFIL(NLEVEL-J+1), 0-based rows
    for term in range(CKD_NAI_MAX):
        prs = (u[lo, 1] + u[hi, 1]) / 2.
        tmp = (u[lo, 2] + u[hi, 2]) / 2.
        conc = (u[lo, 3] + u[hi, 3]) / 2.
        conc = conc * 1.e-06
        for k in range(NBABS):
            if term >= prep["nexp"][k]:
                continue
            xk[k, term], prs, tmp, conc = coeff_abs_ckd_rows(k + 1, prep["ki"][k][term], prep["tab_pres"], prep["tab_temp"],
                                                             prep["tab_conc"], prs, tmp, conc)
    ro = prep["ro"][:, nl - j - 1].copy()             # RO(K, NLEVEL-J)
    return xk, ro


class _LayerGeometry:
    """Everything of COEFF_ABS_CKD that depends on the atmosphere and the table axes only, for the 49 layers."""


@functools.lru_cache(maxsize=8)
def _layer_geometry(u_bytes, pres_bytes, temp_bytes, conc_bytes):
    """Brackets and weights of the three interpolations of COEFF_ABS_CKD (src/SOS_SUB_TRS.F:171) for the 49 layers of an
    atmosphere on given table axes.  The state the reference carries from gas to gas (pressure, temperature and water-vapour
    concentration clamped to the tables) is the same clamp for every gas -- idempotent -- and every interpolation works on the
    clamped values: the geometry is one per atmosphere, whatever gases and terms are present."""
    u = np.frombuffer(u_bytes, dtype=np.float64).reshape(NLEVEL, -1)
    tab_pres, x, tab_conc = (np.frombuffer(b, dtype=np.float64) for b in (pres_bytes, temp_bytes, conc_bytes))
    nl = NLEVEL
    j = np.arange(1, nl)
    lo, hi = nl - j - 1, nl - j
    prs = (u[lo, 1] + u[hi, 1]) / 2.
    tmp = (u[lo, 2] + u[hi, 2]) / 2.
    conc = (u[lo, 3] + u[hi, 3]) / 2.
    conc = conc * 1.e-06
    g = _LayerGeometry()
    tmp = np.minimum(np.maximum(tmp, x[0]), x[-1])
    g.act = ~(prs <= tab_pres[0])
    g.n = int(g.act.sum())
    if not g.n:
        return g
    g.pa = pa = np.minimum(prs[g.act], tab_pres[-1])
    g.ta = ta = tmp[g.act]
    g.ca = ca = np.minimum(np.maximum(conc[g.act], tab_conc[0]), tab_conc[-1])
    g.ip = ip = _bracket(tab_pres, pa)
    g.ic = ic = _bracket(tab_conc, ca)
    g.c0, g.c1 = tab_conc[ic][:, None, None], tab_conc[ic + 1][:, None, None]
    g.cd = ca[:, None, None] - g.c1
    g.p1 = tab_pres[ip + 1][:, None]
    g.pw = g.p1 - tab_pres[ip][:, None]
    g.pd = pa[:, None] - g.p1
    nt = len(x)
    klo = np.zeros(g.n, dtype=np.int64); khi = np.full(g.n, nt - 1, dtype=np.int64)
    while np.any(khi - klo > 1):                       # _splint's bisection, as in coeff_abs_ckd_rows
        k = (khi + klo + 2) // 2 - 1
        go = khi - klo > 1
        up = x[k] > ta
        khi = np.where(go & up, k, khi)
        klo = np.where(go & ~up, k, klo)
    g.klo, g.khi, g.rows = klo, khi, np.arange(g.n)
    g.h = h = x[khi] - x[klo]
    g.bad_h = bool(np.any(h == 0.))
    if not g.bad_h:
        g.aa = aa = (x[khi] - ta) / h
        g.bb = bb = (ta - x[klo]) / h
        g.a3, g.b3, g.h2 = aa * aa * aa - aa, bb * bb * bb - bb, h * h
    g.it = it = _bracket(x, ta)
    g.lw = x[it + 1] - x[it]
    g.ld = ta - x[it + 1]
    # the forward sweep of the spline's tridiagonal solve for a finite first slope: SIG, P and the factor D2(K) depend on the
    # temperatures alone (the scalars the row form computes per row)
    sig, pk, dk = np.zeros(nt), np.zeros(nt), np.zeros(nt)
    dk[0] = -0.5
    for k in range(1, nt - 1):
        sig[k] = (x[k] - x[k - 1]) / (x[k + 1] - x[k - 1])
        pk[k] = sig[k] * dk[k - 1] + 2.
        dk[k] = (sig[k] - 1.) / pk[k]
    g.sig, g.pk, g.dk = sig, pk, dk
    g.dx = x[1:] - x[:-1]                              # X(K+1) - X(K)
    g.dx2 = x[2:] - x[:-2]                             # X(K+1) - X(K-1)
    return g


def _spline_rows_finite(g, x, y, dy1, dyn):
    """_spline_rows for finite end slopes, with the x-only quantities of the forward sweep taken from the geometry: the same
    statements on the same values, element by element."""
    n = len(x)
    d2 = np.empty_like(y)
    u = np.empty_like(y)
    u[:, 0] = (3. / (x[1] - x[0])) * ((y[:, 1] - y[:, 0]) / (x[1] - x[0]) - dy1)
    sl = (y[:, 1:] - y[:, :-1]) / g.dx                                  # (Y(K+1) - Y(K)) / (X(K+1) - X(K)), K = 0 .. n-2
    t = 6. * (sl[:, 1:] - sl[:, :-1]) / g.dx2                           # column K-1 holds the term of row K = 1 .. n-2
    for k in range(1, n - 1):
        u[:, k] = (t[:, k - 1] - g.sig[k] * u[:, k - 1]) / g.pk[k]
    un = (3. / (x[n - 1] - x[n - 2])) * (dyn - (y[:, n - 1] - y[:, n - 2]) / (x[n - 1] - x[n - 2]))
    d2[:, n - 1] = (un - 0.5 * u[:, n - 2]) / (0.5 * g.dk[n - 2] + 1.)
    for k in range(n - 2, -1, -1):
        d2[:, k] = g.dk[k] * d2[:, k + 1] + u[:, k]
    return d2


def _pairs_values(g, x, a0, a1):
    """The pressure and temperature interpolations of COEFF_ABS_CKD for a stack of (gas, term) tables already reduced to the two
    bracketing pressure rows: a0, a1 [pairs][La][NT] -> k of every pair and active layer [pairs][La]."""
    if g.bad_h:
        raise AbsorptionError("ERROR for SPLINT interpolation")
    npair, la = a0.shape[0], g.n
    xki = (((a1 - a0) / g.pw) * g.pd + a1).reshape(npair * la, -1)
    dy1 = (xki[:, 1] - xki[:, 0]) / (x[1] - x[0])
    dyn = (xki[:, -1] - xki[:, -2]) / (x[-1] - x[-2])
    if np.any(dy1 > _F(.99E30)) or np.any(dyn > _F(.99E30)):
        d2 = _spline_rows(x, xki, dy1, dyn)
    else:
        d2 = _spline_rows_finite(g, x, xki, dy1, dyn)
    rows = np.arange(npair * la)
    klo, khi = np.tile(g.klo, npair), np.tile(g.khi, npair)
    tl = lambda a: np.tile(a, npair)
    val = tl(g.aa) * xki[rows, klo] + tl(g.bb) * xki[rows, khi] + (tl(g.a3) * d2[rows, klo] + tl(g.b3) * d2[rows, khi]) * tl(g.h2) / 6.
    neg = val < 0.
    if neg.any():
        it = tl(g.it)
        lin = ((xki[rows, it + 1] - xki[rows, it]) / tl(g.lw)) * tl(g.ld) + xki[rows, it + 1]
        val = np.where(neg, lin, val)
        if np.any(val < 0.):
            raise AbsorptionError("COEFF_ABS_CKD : ERROR_923 : Calculations give ki < 0 : uncorrect value!")
    return val.reshape(npair, la)


def _geometry_of(prep):
    return _layer_geometry(np.ascontiguousarray(prep["userprofil"]).tobytes(), np.ascontiguousarray(prep["tab_pres"]).tobytes(),
                           np.ascontiguousarray(prep["tab_temp"]).tobytes(), np.ascontiguousarray(prep["tab_conc"]).tobytes())


def _absorbing_pairs(prep):
    """(gas, term) pairs of the interval whose table is not all zero (NMAXAI = 0: the gas does not absorb here, every
    interpolation of zeros is zero), in the reference's loop order (term outermost)."""
    out = []
    for term in range(CKD_NAI_MAX):
        for k in range(NBABS):
            if term < prep["nexp"][k] and prep["ki"][k][term].any():
                out.append((k, term))
    return out


def layer_tables(prep):
    """XK(gas, term, layer) RO(gas, layer) of SOS_ABSPROFILE.F:325-353 for every exponential term of every gas:
    returns (xk [8][5][49], ro [8][49]) with layer index J-1, J = 1 the TOP layer (the reference's loop order).
    The 49 layers of ALL (term, gas) pairs with a non-zero table are interpolated together, on brackets and weights kept per
    atmosphere (_layer_geometry: a spectrum shares one atmosphere); element by element the arithmetic of coeff_abs_ckd --
    tests/test_absorption.py holds it against the pair-by-pair and the layer-by-layer forms, bit for bit.  A prep that went
    through layer_tables_many carries its tables already."""
    done = prep.get("_layer_tables")
    if done is not None:
        return done
    nl = NLEVEL
    xk = np.zeros((NBABS, CKD_NAI_MAX, nl - 1))
    j = np.arange(1, nl)
    ro = prep["ro"][:, nl - j - 1].copy()             # RO(K, NLEVEL-J)
    g = _geometry_of(prep)
    pairs = _absorbing_pairs(prep) if g.n else []
    if pairs:
        a0s, a1s = [], []
        for k, term in pairs:
            ki = prep["ki"][k][term]
            if k == 0:                                 # water vapour: first along the concentration axis
                k0, k1 = ki[g.ic], ki[g.ic + 1]                                # [La][NP][NT]
                xkh = ((k1 - k0) / (g.c1 - g.c0)) * g.cd + k1
                a0, a1 = xkh[g.rows, g.ip], xkh[g.rows, g.ip + 1]
            else:
                a0, a1 = ki[g.ip], ki[g.ip + 1]                                # [La][NT]
            a0s.append(a0)
            a1s.append(a1)
        val = _pairs_values(g, prep["tab_temp"], np.stack(a0s), np.stack(a1s))
        for q, (k, term) in enumerate(pairs):
            xk[k, term][g.act] = val[q]
    return xk, ro


def layer_tables_many(preps):
    """layer_tables for the wavelengths of a spectrum in one pass: the (gas, term) pairs of ALL intervals that share an
    atmosphere and table axes go through the interpolations together (the same element-wise statements on longer arrays:
    bit-identical to the one-by-one form).  Attaches the tables to every prep (layer_tables then returns them)."""
    nl = NLEVEL
    j = np.arange(1, nl)
    groups = {}
    for prep in preps:
        if prep.get("_layer_tables") is not None:
            continue
        g = _geometry_of(prep)
        groups.setdefault(id(g), (g, []))[1].append(prep)
    for g, members in groups.values():
        outs = [np.zeros((NBABS, CKD_NAI_MAX, nl - 1)) for _ in members]
        if g.n:
            where, dry, wet = [], [], []                 # (member, gas, term) of every pair; tables of the dry gases / of H2O
            for m, prep in enumerate(members):
                for k, term in _absorbing_pairs(prep):
                    (wet if k == 0 else dry).append((m, k, term))
            a0s, a1s = [], []
            x = members[0]["tab_temp"]
            if dry:
                kis = np.stack([members[m]["ki"][k][term] for m, k, term in dry])          # [P][NP][NT]
                a0s.append(kis[:, g.ip]); a1s.append(kis[:, g.ip + 1])                      # [P][La][NT]
                where += dry
            if wet:
                kis = np.stack([members[m]["ki"][k][term] for m, k, term in wet])          # [P][NC][NP][NT]
                k0, k1 = kis[:, g.ic], kis[:, g.ic + 1]                                    # [P][La][NP][NT]
                xkh = ((k1 - k0) / (g.c1 - g.c0)) * g.cd + k1
                a0s.append(xkh[:, g.rows, g.ip]); a1s.append(xkh[:, g.rows, g.ip + 1])
                where += wet
            if where:
                val = _pairs_values(g, x, np.concatenate(a0s), np.concatenate(a1s))
                for q, (m, k, term) in enumerate(where):
                    outs[m][k, term][g.act] = val[q]
        for prep, xk in zip(members, outs):
            prep["_layer_tables"] = (xk, prep["ro"][:, nl - j - 1].copy())


def layer_tables_scalar(prep):
    """The layer-by-layer form of layer_tables (one coeff_abs_ckd call per layer, term and gas): kept as the statement-for-
    statement restatement the vectorised form is tested against."""
    u = prep["userprofil"]
    nl = NLEVEL
    xk = np.zeros((NBABS, CKD_NAI_MAX, nl - 1))
    ro = np.zeros((NBABS, nl - 1))
    for j in range(1, nl):
        lo, hi = nl - j - 1, nl - j
        for term in range(CKD_NAI_MAX):
            prs = (u[lo, 1] + u[hi, 1]) / 2.
            tmp = (u[lo, 2] + u[hi, 2]) / 2.
            conc = (u[lo, 3] + u[hi, 3]) / 2.
            conc = conc * 1.e-06
            for k in range(NBABS):
                if term >= prep["nexp"][k]:
                    continue
                xk[k, term, j - 1], prs, tmp, conc = coeff_abs_ckd(k + 1, prep["ki"][k][term], prep["tab_pres"], prep["tab_temp"],
                                                                   prep["tab_conc"], prs, tmp, conc)
        ro[:, j - 1] = prep["ro"][:, nl - j - 1]
    return xk, ro


def absprofile_host(xk, ro, ik):
    """SOS_ABSPROFILE.F:325-371 for one bin: ik[8] 1-based term per gas -> TAUABS[50] (level 1 = TOA)."""
    nl = xk.shape[2] + 1
    tau = np.zeros(nl)
    trs = 1.0
    for j in range(nl - 1):
        t1c = 0.
        for k in range(NBABS):
            t1c = t1c + xk[k, ik[k] - 1, j] * ro[k, j]
        trs = trs * np.exp(-t1c)
        tau[j + 1] = -np.log(trs) if trs > 0. else TAUABS_MAX
    return tau


def band_bin_count(wa, nustep, root=None):
    """Number of CKD bins of the spectral interval holding wavelength `wa` (microns): the product of the gases' numbers of
    exponential terms NEXP (SOS_PROC.F:3381-3404), from the table files alone."""
    nu = _F(1.0E+4) / wa
    if nu > CKD_NUMAX or nu < CKD_NUMIN:
        raise AbsorptionError("The simulation wavelength is not included in the spectral range of CKD data")
    nb = 1
    for k in range(1, NBABS + 1):
        g = read_ckd_coeff(k, nu, nustep, root)
        nb *= int(g["nexp"][int((g["numax"] - nu) / nustep)])
    return nb


def bin_costs(ik, xk, ro, tau_scat):
    """Relative solver cost of every CKD bin of a band (dist.bin_cost) from its total gas absorption optical depth
    sum_gas sum_layer k(gas, term, layer) RO(gas, layer) (the end value of SOS_ABSPROFILE.F:325-371 before the clamp)."""
    from .dist import bin_cost
    ik = np.asarray(ik)
    col = np.einsum("gtl,gl->gt", xk, ro)                   # [gas][term] column optical depth
    tg = col[np.arange(ik.shape[1])[None, :], ik - 1].sum(axis=1)
    return bin_cost(tau_scat, tg)


def bins(prep):
    """Bin list of the wavelength in the reference's solve order: (ik[nb][8] 1-based, aik[nb] normalised, sum before
    normalisation).  Raises like SOS_PROC.F:3414 when the weights do not sum to 1 within 1e-6."""
    ik, aik, s = ckd_bin_weights(prep["nexp"], prep["kdis_ai"])
    if abs(s - 1.0) >= 1.e-06:
        raise AbsorptionError("sum of the CKD weights AIK = %r differs from 1 (SOS_PROC ERROR_3500)" % s)
    return ik, aik, s
