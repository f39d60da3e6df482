"""Seeded synthetic inputs for the hot path: angles, phase-matrix
coefficients, per-CKD-bin atmospheric profiles and bin weights.

These are INPUT generators only (no solver arithmetic): the same arrays are fed, byte for byte, to the
HIP path, to the oracle and to the reference Fortran.  Values are quantised the way the reference's
file hand-offs quantise them (angles D21.14 -- SOS_ANGLES.F:647, profile E15.8 -- SOS.F:692,
phase coefficients E15.8 -- SOS_AEROSOLS.F:3056) so that "identical inputs" means the arrays SOS_OS
really sees.
"""
import numpy as np

MDF_DEFAULT = float(np.float32(0.0279))  # RON=CTE_MDF is a REAL*4 literal widened to double (SOS.h:373, SOS_PROC.F:1289)


def _round_sig(x, sig):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    nz = x != 0
    mag = np.floor(np.log10(np.abs(x[nz])))
    scale = 10.0 ** (sig - 1 - mag)
    out[nz] = np.round(x[nz] * scale) / scale
    return out


def gauss_angles(ng, tetas_deg):
    """Positive half of the 2*ng-point Gauss-Legendre rule, mu descending, with the solar direction
    inserted with weight 0 (SOS_ANGLES.F:401-466, 837-852).  Returns (mu[N], w[N], n0) with n0 the
    1-based index of mus (N0 of SOS_OS)."""
    x, w = np.polynomial.legendre.leggauss(2 * ng)
    mu = x[ng:][::-1].copy()
    wt = w[ng:][::-1].copy()
    mus = np.cos(np.deg2rad(tetas_deg))
    hit = np.where(np.abs(mus - mu) < 1e-5)[0]  # CTE_SEUIL_ECART_MUS, SOS.h:561
    if len(hit):
        n0 = int(hit[-1]) + 1
    else:
        pos = int(np.sum(mu > mus))
        mu = np.insert(mu, pos, mus)
        wt = np.insert(wt, pos, 0.0)
        n0 = pos + 1
    # D21.14 in SOS_UsedAngles.txt (SOS_ANGLES.F:647) keeps 14 significant digits
    return _round_sig(mu, 14), _round_sig(wt, 14), n0


def hg_phase(os_nb, g, polar=(0.9, 0.85, -0.08)):
    """Synthetic Henyey-Greenstein-like expansion: beta_l=(2l+1)g^l, alpha_l=a*beta_l, zeta_l=z*beta_l,
    gamma_l=c*beta_l for l>=2 (0 for l<2).  Returns alpha,beta,gamma,zeta (os_nb+1 each)."""
    l = np.arange(os_nb + 1)
    beta = (2 * l + 1) * g ** l
    a, z, c = polar
    m = (l >= 2).astype(np.float64)
    return (_round_sig(a * beta * m, 8), _round_sig(beta, 8), _round_sig(c * beta * m, 8),
            _round_sig(z * beta * m, 8))


def profile(nt, tau_r=0.0948, tau_a=0.3, k_abs=0.0, hr=8.0, ha=2.0, hg=4.0, ztoa=120.0):
    """Profile(s) on nt layers of equal total optical thickness (the reference places its levels by
    bisection on tau(z) too, SOS_PROFIL.F:1210-1329).  k_abs may be a scalar (one bin) or an array (one bin
    per element, vectorised).  Returns h, xdel, ydel, zprof ([nt+1] or [nb][nt+1]), i.e. the PROFIL_TMP
    columns before SOS.F's truncation rescale."""
    k = np.atleast_1d(np.asarray(k_abs, dtype=np.float64))[:, None]
    def tr(z): return tau_r * np.exp(-z / hr)
    def ta(z): return tau_a * np.exp(-z / ha)
    def tg(z): return k * np.exp(-z / hg)
    def tt(z): return tr(z) + ta(z) + tg(z)
    zt = np.full((k.shape[0], 1), ztoa)
    t0, t1 = tt(zt), tt(np.zeros_like(zt))
    targets = t0 + (t1 - t0) * (np.arange(nt + 1) / nt)[None, :]
    lo = np.zeros_like(targets)
    hi = np.full_like(targets, ztoa)
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        big = tt(mid) > targets
        lo = np.where(big, mid, lo)
        hi = np.where(big, hi, mid)
    z = 0.5 * (lo + hi)
    z[:, 0], z[:, -1] = ztoa, 0.0
    z = np.round(z, 5)  # F10.5
    h = tt(z)
    xdel = np.zeros_like(h)
    ydel = np.zeros_like(h)
    dt = np.diff(h, axis=1)
    xdel[:, 1:] = np.diff(ta(z), axis=1) / dt
    ydel[:, 1:] = np.diff(tr(z), axis=1) / dt
    d0 = tau_r / hr * np.exp(-ztoa / hr) + tau_a / ha * np.exp(-ztoa / ha) + k[:, 0] / hg * np.exp(-ztoa / hg)
    xdel[:, 0] = (tau_a / ha * np.exp(-ztoa / ha)) / d0
    ydel[:, 0] = (tau_r / hr * np.exp(-ztoa / hr)) / d0
    out = (_round_sig(h, 8), _round_sig(xdel, 8), _round_sig(ydel, 8), z)
    if np.ndim(k_abs) == 0:
        return tuple(a[0] for a in out)
    return out


def ckd_bins(nb, nt, seed=1234, tau_r=0.0948, tau_a=0.3, kmin=1e-3, kmax=30.0, part=None):
    """nb per-bin profiles differing by the gas absorption (k_b log-uniform in [kmin,kmax]) and
    Dirichlet(1) weights AIK normalised to 1 (SOS_PROC.F:3481-3487).  Returns dict of arrays
    h/xdel/ydel/zprof [nb][nt+1], aik [nb], k_abs [nb].  part = (lo, hi): only that slice of the nb bins is built (the
    random draws are those of the whole band, so a rank's slice equals the same rows of the full call)."""
    rng = np.random.default_rng(seed)
    k = np.exp(rng.uniform(np.log(kmin), np.log(kmax), nb))
    aik = rng.dirichlet(np.ones(nb)) if nb > 1 else np.ones(1)
    if part is not None:
        k, aik = k[part[0]:part[1]], aik[part[0]:part[1]]
    H, X, Y, Z = profile(nt, tau_r, tau_a, k)
    return dict(h=H, xdel=X, ydel=Y, zprof=Z, aik=aik, k_abs=k)


def rescale_profile(h, xdel, ydel, a_tronc, piz, piztr, os_nb):
    """Host-side restatement of the reference per-bin truncation rescale (SOS.F:523-550): inputs are the
    PROFIL_TMP columns, outputs the arrays SOS_OS receives plus IBORM.  Vectorised over leading dims."""
    h = np.array(h, dtype=np.float64); xdel = np.array(xdel, dtype=np.float64); ydel = np.array(ydel, dtype=np.float64)
    if a_tronc != 0.0:
        dh = np.diff(h, axis=-1)
        va = xdel[..., 1:] * dh
        vatr = va * (1 - piz * 0.5 * a_tronc)
        vr = ydel[..., 1:] * dh
        vg = (1 - xdel[..., 1:] - ydel[..., 1:]) * dh
        tot = vatr + vr + vg
        # the reference accumulates HTR(I) = (..) + HTR(I-1) serially from HTR(0)=H(0); cumsum keeps that order
        htr = np.cumsum(np.concatenate([h[..., :1], tot], axis=-1), axis=-1)
        xdel[..., 1:] = vatr / tot
        ydel[..., 1:] = vr / tot
        h = htr
    xdel = xdel * piztr
    iborm = os_nb if np.any(xdel != 0.0) else 2
    return h, xdel, ydel, iborm
