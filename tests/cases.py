"""Shared definitions of the SOS_OS parity cases (inputs only; used by the golden generator,
the oracle tests and the GPU parity tests)."""
import importlib

import numpy as np

S = importlib.import_module("radiativetransfer-sos_amd.synth")


def _surf_matrices(n, smax, seed):
    """Small synthetic REAL*4 BRDF/BPDF Fourier matrices (decaying with the order) in FICSURF record
    order [s][ab][J][I]; only used to exercise the IMAT_SURF=1 code path."""
    rng = np.random.default_rng(seed)
    r = rng.uniform(-1.0, 1.0, (smax + 1, 9, n, n))
    scale = np.array([0.05, 0.01, 0.01, 0.01, 0.02, 0.005, 0.01, 0.005, 0.02])[None, :, None, None]
    decay = (0.6 ** np.arange(smax + 1))[:, None, None, None]
    r = r * scale * decay
    r[:, 0] = np.abs(r[:, 0])
    return r.astype(np.float32)


def make_case(name):
    """Returns dict(kw for sos_os-style calls): rmu, ga, os_nb, bins (list of (h,xdel,ydel,zprof)), common kwargs."""
    c = dict(name=name)
    if name == "rayleigh_n25":            # BASELINE config 1: Rayleigh only, Lambert, 24 Gauss
        ng, nt, os_nb, g, kabs = 24, 30, 80, None, [0.0]
        kw = dict(ro=0.1)
    elif name == "aer_n41":               # BASELINE config 2: aerosol + Rayleigh, 40 Gauss, 30 layers, Lambert
        ng, nt, os_nb, g, kabs = 40, 30, 80, 0.75, [0.0, 0.5, 5.0]
        kw = dict(ro=0.1)
    elif name == "aer_n41_g09":
        ng, nt, os_nb, g, kabs = 40, 30, 80, 0.9, [1.0]
        kw = dict(ro=0.1)
    elif name == "fresnel_n41":           # config 3 surface: flat sea
        ng, nt, os_nb, g, kabs = 40, 30, 80, 0.75, [0.0, 2.0]
        kw = dict(ro=0.05, ifresnel=1, ind_surf=1.34)
    elif name == "nopolar_n41":
        ng, nt, os_nb, g, kabs = 40, 30, 80, 0.75, [0.3]
        kw = dict(ro=0.1, ipolar=0)
    elif name == "zout_n25_nt60":
        ng, nt, os_nb, g, kabs = 24, 60, 48, 0.8, [0.0, 1.0]
        kw = dict(ro=0.1, zout=3.0)
    elif name == "brdf_n13":              # IMAT_SURF=1 path with small synthetic matrices
        ng, nt, os_nb, g, kabs = 12, 20, 24, 0.6, [0.0, 0.7]
        kw = dict(ro=0.02, imat_surf=1)
    elif name == "brdf_zout_n13":
        ng, nt, os_nb, g, kabs = 12, 20, 24, 0.6, [0.2]
        kw = dict(ro=0.02, imat_surf=1, zout=1.5, ipolar=0)
    elif name == "black_n9":              # ro = 0, tiny
        ng, nt, os_nb, g, kabs = 8, 12, 16, 0.5, [0.0]
        kw = dict(ro=0.0)
    elif name == "igmax_n9":              # IGMAX reached
        ng, nt, os_nb, g, kabs = 8, 12, 16, 0.5, [0.0]
        kw = dict(ro=0.3, igmax=4)
    # ---- NT beyond the LDS-resident variants: field-in-HBM kernel (reference profiles have NT 100..600) ----
    elif name == "rayleigh_n25_nt101":    # config 1 with the level count SOS_PROFILE really produces
        ng, nt, os_nb, g, kabs = 24, 101, 80, None, [0.0]
        kw = dict(ro=0.1)
    elif name == "aer_n41_nt120":
        ng, nt, os_nb, g, kabs = 40, 120, 80, 0.75, [0.0, 3.0]
        kw = dict(ro=0.1)
    elif name == "fresnel_zout_n25_nt70":
        ng, nt, os_nb, g, kabs = 24, 70, 48, 0.7, [0.5]
        kw = dict(ro=0.03, ifresnel=1, ind_surf=1.34, zout=2.0)
    elif name == "brdf_n13_nt97":
        ng, nt, os_nb, g, kabs = 12, 97, 24, 0.6, [0.3]
        kw = dict(ro=0.02, imat_surf=1, zout=0.7)
    elif name == "aer_n9_nt600":          # CTE_OS_NT = 600 (SOS.h:202)
        ng, nt, os_nb, g, kabs = 8, 600, 16, 0.5, [1.0]
        kw = dict(ro=0.1)
    elif name == "brdf_zout_n30_nt110":   # streamed field of six row tiles per half (N = 27 ... 32)
        ng, nt, os_nb, g, kabs = 29, 110, 48, 0.7, [0.2]
        kw = dict(ro=0.02, imat_surf=1, zout=1.2)
    elif name == "aer_n32_nt75":          # ... at its last direction count (96 rows)
        ng, nt, os_nb, g, kabs = 31, 75, 48, 0.75, [0.0, 1.0]
        kw = dict(ro=0.1)
    elif name == "aer_n26_nt90":          # five row tiles per half at its last direction count (78 rows)
        ng, nt, os_nb, g, kabs = 25, 90, 48, 0.75, [0.6]
        kw = dict(ro=0.1)
    elif name == "aer_n21":               # NW=4, one row tile per wave
        ng, nt, os_nb, g, kabs = 20, 28, 40, 0.7, [0.0, 1.5]
        kw = dict(ro=0.15)
    elif name == "zout_n21_nt50":         # one row tile per wave, four column tiles
        ng, nt, os_nb, g, kabs = 20, 50, 40, 0.7, [0.4]
        kw = dict(ro=0.05, zout=4.0)
    elif name == "aer_n49":               # N > 42: eight-wave workgroups
        ng, nt, os_nb, g, kabs = 48, 26, 80, 0.75, [0.0, 2.0]
        kw = dict(ro=0.1)
    elif name == "fresnel_n80_nt24":      # CTE_OS_NBMU_MAX = 80 directions (SOS.h:471)
        ng, nt, os_nb, g, kabs = 79, 24, 80, 0.8, [0.3]
        kw = dict(ro=0.03, ifresnel=1, ind_surf=1.34)
    elif name == "zout_n65_nt45":         # N > 42 with the field in the HBM scratch
        ng, nt, os_nb, g, kabs = 64, 45, 64, 0.7, [0.6]
        kw = dict(ro=0.2, zout=2.5)
    elif name == "user_angles_n28":       # 24 Gauss + sun + 3 user viewing angles (zero quadrature weight, SOS_ANGLES)
        ng, nt, os_nb, g, kabs = 24, 30, 48, 0.7, [0.0, 1.0]
        kw = dict(ro=0.1)
    elif name.startswith("x_"):           # ad-hoc debugging case: x_<ng>_<nt>_<os_nb>_<g>_<zout or -1>
        f = name.split("_")
        ng, nt, os_nb, g, kabs = int(f[1]), int(f[2]), int(f[3]), float(f[4]), [0.0]
        kw = dict(ro=0.1)
        if float(f[5]) >= 0:
            kw["zout"] = float(f[5])
    else:
        raise KeyError(name)
    mu, w, n0 = S.gauss_angles(ng, 35.0)
    if name == "user_angles_n28":
        # user angles are merged into the descending list with weight 0, exactly like the solar direction
        extra = np.cos(np.radians([10.0, 47.5, 72.0]))
        mu_all = np.concatenate([mu, extra]); w_all = np.concatenate([w, np.zeros(3)])
        order = np.argsort(-mu_all, kind="stable")
        mus = mu[n0 - 1]
        mu, w = mu_all[order], w_all[order]
        n0 = int(np.where(mu == mus)[0][0]) + 1
    if g is None:
        al, be, ga, ze = S.hg_phase(os_nb, 0.0)
    else:
        al, be, ga, ze = S.hg_phase(os_nb, g)
    bins = []
    iborm = os_nb
    for k in kabs:
        h, x, y, z = S.profile(nt, k_abs=k)
        if g is None:
            x = np.zeros_like(x)
        h, x, y, ib = S.rescale_profile(h, x, y, 0.0, 0.95, 0.95, os_nb)
        iborm = ib
        bins.append((h, x, y, z))
    if kw.get("imat_surf"):
        kw["rsurf"] = _surf_matrices(len(mu), iborm, 7)
    c.update(rmu=mu, ga=w, n0=n0, os_nb=os_nb, coefs=(al, be, ga, ze), bins=bins, iborm=iborm, kw=kw)
    return c


ALL_CASES = ["rayleigh_n25", "aer_n41", "aer_n41_g09", "fresnel_n41", "nopolar_n41", "zout_n25_nt60",
             "brdf_n13", "brdf_zout_n13", "black_n9", "igmax_n9",
             "rayleigh_n25_nt101", "aer_n41_nt120", "fresnel_zout_n25_nt70", "brdf_n13_nt97", "aer_n9_nt600",
             "aer_n21", "zout_n21_nt50", "aer_n49", "fresnel_n80_nt24", "zout_n65_nt45", "user_angles_n28",
             "brdf_zout_n30_nt110", "aer_n32_nt75", "aer_n26_nt90"]


def run_cpu(mod, case, b):
    """mod = oracle.oracle_ctypes or oracle.ref_ctypes."""
    h, x, y, z = case["bins"][b]
    al, be, ga, ze = case["coefs"]
    return mod.sos_os(case["rmu"], case["ga"], case["os_nb"], h, x, y, al, be, ga, ze, n0=case["n0"], zprof=z,
                      iborm=case["iborm"], **case["kw"])


def run_gpu(pkg, case):
    """All bins of the case in one batch through the C ABI.  Returns list of dicts like run_cpu."""
    import torch
    kw = dict(case["kw"])
    zout = kw.pop("zout", -1.0)
    al, be, ga, ze = case["coefs"]
    cx = pkg.SosContext(case["rmu"], case["ga"], case["n0"], al, be, ga, ze, iborm_max=case["iborm"], **kw)
    H = np.array([b[0] for b in case["bins"]]); X = np.array([b[1] for b in case["bins"]])
    Y = np.array([b[2] for b in case["bins"]]); Z = np.array([b[3] for b in case["bins"]])
    bins = cx.upload_bins(H, X, Y, iborm=np.full(len(H), case["iborm"], dtype=np.int32), zout=zout, zprof=Z)
    out = cx.solve(bins)
    torch.cuda.synchronize()
    res = []
    for b in range(len(H)):
        f = int(out["norders"][b])
        res.append(dict(records=out["rec"][b, :f].cpu().numpy(), ig_counts=out["iglast"][b, :f].cpu().numpy(),
                        emoins=float(out["flux"][b, 0]), eplus=float(out["flux"][b, 1])))
    cx.close()
    return res


def compare_records(a, b, rtol=1e-9, what=""):
    """Parity bar: |a-b| <= rtol*|b| + rtol*1e-3*max|I| (the absolute floor covers Q/U entries that are
    sums cancelling to ~0, e.g. U at the principal plane; it is 1e-12 of the I scale)."""
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = np.abs(b[:, 0]).max() if b.size else 1.0
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + rtol * 1e-3 * scale
    bad = err > tol
    assert not bad.any(), "%s: %d entries exceed tolerance; worst err/tol %.3g" % (what, bad.sum(), (err / tol).max())
    return float((err / np.maximum(np.abs(b), 1e-3 * scale)).max())


# ---- SOS_PROFILE cases: name -> (tr, hr, ta, ha, gas scale k or None, gas scale height) -------------
PROFILE_CASES = {
    "nogas": (0.0948, 8.0, 0.3, 2.0, None, 7.0),
    "ray_only": (0.0948, 8.0, 0.0, 2.0, None, 7.0),
    "thin": (0.01, 8.0, 0.005, 2.0, None, 7.0),
    "thick_aer": (0.0948, 8.0, 1.5, 1.5, None, 7.0),
    "gas_weak": (0.0948, 8.0, 0.3, 2.0, 0.4, 7.0),
    "gas_mid": (0.0948, 8.0, 0.3, 2.0, 1.2, 7.0),
    "gas_strong": (0.0948, 8.0, 0.3, 2.0, 8.0, 7.0),
    "gas_very_strong": (0.0948, 8.0, 0.1, 3.0, 60.0, 5.0),
    "gas_ray": (0.0948, 8.0, 0.0, 2.0, 0.7, 7.0),
    "gas_thin": (0.004, 8.0, 0.002, 2.0, 0.003, 7.0),
    "gas_h2o_like": (0.02, 8.0, 0.15, 2.0, 0.9, 2.0),
}


def profile_case(name):
    """Inputs of one SOS_PROFILE case: dict(tr, hr, ta, ha, altabs[50] | None, tabs[50] | None).  The absorption profile is
    a synthetic exponential column sampled on 50 descending altitudes (CTE_ABS_NBLEV levels from 120 km to the ground,
    zero at the top level like the reference's profiles)."""
    tr, hr, ta, ha, k, hg = PROFILE_CASES[name]
    if k is None:
        return dict(tr=tr, hr=hr, ta=ta, ha=ha, altabs=None, tabs=None)
    alt = np.concatenate([np.linspace(120.0, 30.0, 10), np.linspace(28.0, 0.0, 40)])
    tab = k * np.exp(-alt / hg)
    tab[0] = 0.0
    return dict(tr=tr, hr=hr, ta=ta, ha=ha, altabs=alt, tabs=tab)


def compare_proc_outputs(rs, out, g, coef_tronca=None, rtol=1e-9):
    """run_sos.sos_proc 23-tuple against a reference SOS_PROC golden (tests/golden/sos_proc_*.npz): I,Q,U tables to 1e-9
    relative (plus 1e-12 of the I scale for near-zero Q/U), angles / flux scalars to 1e-9; identical table shapes and fill
    pattern.  coef_tronca: expected value of the last output when it differs from the golden's own run."""
    assert len(out) == 23
    n = int(g["nblum"])
    assert out[0] == n and np.array_equal(out[1], g["ind_angout"])
    assert out[2].shape == (361,) and out[3].shape == (81,) and out[5].shape == (361, 81)
    assert np.allclose(out[2], g["phi"], atol=1e-12) and np.allclose(out[3], g["vza"], atol=1e-10)
    scale = np.abs(g["i_up"]).max()
    for k, nm in enumerate(rs.OUTPUT_NAMES):
        if k < 4:
            continue
        exp = g[nm]
        got = np.asarray(out[k])
        if nm.startswith(("i_", "q_", "u_", "l_pol")):
            tol = rtol * np.abs(exp) + 1e-3 * rtol * scale
            assert np.all(np.abs(got - exp) <= tol), (nm, np.abs(got - exp).max())
        elif nm.startswith("sca_ang"):
            # acos is ill-conditioned at exact forward/backward scattering: compare cosines tightly, angles loosely
            assert np.allclose(np.cos(np.radians(got)), np.cos(np.radians(exp)), rtol=0, atol=1e-13), nm
            assert np.allclose(got, exp, rtol=0, atol=1e-5), nm
        elif nm.startswith(("pol_ang", "pol_rate")):
            # angle/rate of polarisation are ill-conditioned where Q,U ~ 0: compare where the polarised radiance is significant
            lp = g["l_pol_up"] if nm.endswith("up") else g["l_pol_down"]
            m = lp > 1e-6 * scale
            assert np.allclose(got[m], exp[m], rtol=1e-6, atol=1e-6), nm
            assert np.array_equal(got == -999.0, exp == -999.0) or not ((got == -999.0) ^ (exp == -999.0))[m].any()
        else:
            if nm == "coef_tronca" and coef_tronca is not None:
                exp = coef_tronca
            assert abs(got - exp) <= rtol * abs(exp) + 1e-15, (nm, got, exp)
