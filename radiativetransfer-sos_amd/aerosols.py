"""Aerosol optical properties of one wavelength: what the reference's SOS_AEROSOLS writes to Aerosols.txt
for the size-distribution models -- mono-modal log-normal / Junge (`-AER.Model 0`) and bimodal log-normal (`-AER.Model 3`).

    mie_angles        <- SOS_ANGLES for the Mie angle set  src/SOS_ANGLES.F:380-466 (Gauss nodes, D21.14 text values)
    alpha_grid        <- the size-parameter loop of SOS_MIE  src/SOS_MIE.F:434-443, 707-708
    mie_records       <- SOS_MIE + SOS_FPHASE_MIE on the GPU (csrc/mie.hip, C ABI sosgpu_mie), no MIE cache file
    granu             <- SOS_GRANU  src/SOS_AEROSOLS.F:4392-4820  size-distribution integral of the Mie records
    decompo_legendre  <- SOS_DECOMPO_LEGENDRE  src/SOS_AEROSOLS.F:3924-4390  truncation + Legendre expansions
    aerosols          <- SOS_AEROSOLS  src/SOS_AEROSOLS.F:680 (IMOD = 0: :1150-1290; IMOD = 3: :1710-2125; closing :2771-2890)

The other models (WMO, Shettle & Fenn, external phase functions, user mixtures) are not built; `-AER.UserFile` covers them.
REAL*4 variables and literals of the Fortran are kept REAL*4 (`np.float32`) where they decide a value."""
import ctypes as C
import math

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


def mie_angles(nb_gauss):
    """Mie angle set without user angles: positive Gauss nodes of the 2 nb_gauss-point rule, ascending, as re-read from
    Aer_UsedAngles.txt (D21.14).  Returns xmu[-N:N], xhr[-N:N] as arrays of 2N+1 (index j + N; entry N unused = 0)."""
    x, w = np.polynomial.legendre.leggauss(2 * nb_gauss)
    mu, wt = _round_sig(x[nb_gauss:], 14), _round_sig(w[nb_gauss:], 14)
    n = nb_gauss
    xmu, xhr = np.zeros(2 * n + 1), np.zeros(2 * n + 1)
    xmu[n + 1:], xhr[n + 1:] = mu, wt
    xmu[:n], xhr[:n] = -mu[::-1], wt[::-1]
    return xmu, xhr


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


def mie_records(xmu, rn, in_, alphao, alphaf, device=0):
    """The records of the reference's MIE file for (rn, in_) on the grid alpha_grid(alphao, alphaf):
    dict(alpha, qext, qsca float32 [na]; g float64 [na]; imie, qmie, umie float32 [na][2N+1])."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("mie_records needs a GPU (gfx950); there is no CPU fallback in the product path")
    al = alpha_grid(alphao, alphaf)
    xmu = np.ascontiguousarray(xmu, dtype=np.float64)
    w = len(xmu)
    nbmu = (w - 1) // 2
    dev = torch.device("cuda", device)
    rec = torch.zeros((len(al), 4 + 3 * w), dtype=torch.float32, device=dev)
    g = torch.zeros(len(al), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = capi.lib().sosgpu_mie(device, nbmu, xmu.ctypes.data_as(C.c_void_p), float(rn), float(in_), len(al),
                               al.ctypes.data_as(C.c_void_p), C.c_void_p(rec.data_ptr()), C.c_void_p(g.data_ptr()), st)
    if rc == -3:
        raise AerosolError("size parameter up to %g: more Mie coefficients than the device kernel holds" % alphaf)
    capi.check(rc, "sosgpu_mie")
    r = rec.cpu().numpy()
    return dict(alpha=r[:, 0].copy(), qext=r[:, 1].copy(), qsca=r[:, 2].copy(), g=g.cpu().numpy(), imie=r[:, 4:4 + w].copy(),
                qmie=r[:, 4 + w:4 + 2 * w].copy(), umie=r[:, 4 + 2 * w:4 + 3 * w].copy(), alphaf=float(alphaf))


def granu(rec, igranu, v1, v2, v3, wa):
    """SOS_GRANU: integral of the Mie records over the size distribution (igranu 1: log-normal, modal radius v1, ln-std v2;
    2: Junge, r0 = v1, slope v2, rmax = v3).  Returns kmat1, kmat2 (per particle), somme_nr, p11, p12, p33 [2N+1]."""
    alpha = rec["alpha"]                               # REAL*4 in the file
    a64 = alpha.astype(np.float64)
    r = a64 * wa / 2. / math.pi
    pas = np.full(len(alpha), np.float32(0.0001), dtype=np.float32)
    for lim, st in ((0.10, 0.001), (1.00, 0.01), (10., 0.05), (30., 0.10), (100., 1.00)):
        pas[alpha > np.float32(lim)] = np.float32(st)
    pas_prev = np.concatenate([[np.float32(0.0001)], pas[:-1]]).astype(np.float64)
    stop = a64 >= (rec["alphaf"] - pas_prev)
    nuse = int(np.argmax(stop)) if stop.any() else len(alpha)
    if igranu == 1:
        b = np.log(r / v1) / v2
        nr = np.exp(-b * b / 2.) / (r * v2 * math.sqrt(2 * math.pi))
    elif igranu == 2:
        over = r > v3
        if over[:nuse].any():
            nuse = int(np.argmax(over))
        nr = np.where(r <= v1, v1 ** (-v2), r ** (-v2))
    else:
        raise AerosolError("unknown size distribution %d" % igranu)
    sl = slice(0, nuse)
    pr = wa * pas[sl].astype(np.float64) / 2. / math.pi
    x1 = nr[sl] * pr * math.pi * r[sl] ** 2
    kmat1 = float(np.sum(x1 * rec["qext"][sl].astype(np.float64)))
    x1s = rec["qsca"][sl].astype(np.float64) * x1
    kmat2 = float(np.sum(x1s))
    p11 = (rec["imie"][sl].astype(np.float64) * x1s[:, None]).sum(0) / kmat2
    p12 = (rec["qmie"][sl].astype(np.float64) * x1s[:, None]).sum(0) / kmat2
    p33 = (rec["umie"][sl].astype(np.float64) * x1s[:, None]).sum(0) / kmat2
    somme_nr = float(np.sum(nr[sl] * pr))
    return kmat1 / somme_nr, kmat2 / somme_nr, somme_nr, p11, p12, p33


def decompo_legendre(itronc, xmu, xhr, os_nb, p11_in, p12, p22, p33):
    """SOS_DECOMPO_LEGENDRE: forward-peak truncation (log-linear extrapolation of P11 beyond acos(0.94), slope taken between
    acos(0.8) and acos(0.94)) and the expansions alpha, beta, gamma, zeta (0:os_nb), normalised by beta_0.
    Returns dict(alpha, beta, gamma, zeta, beta22, delta33, coef_tronca, itronc)."""
    w = len(xmu)
    n = (w - 1) // 2
    J = lambda j: j + n
    ttt = p11_in.copy()
    kk = np.arange(os_nb + 1)
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
        beta11 = np.zeros(os_nb + 1)
        for j in range(-n, n + 1):
            if j == 0:
                continue
            x, xr = p11[J(j)] * xhr[J(j)], xmu[J(j)]
            plm, pl = 0., 1.
            for k in range(os_nb + 1):
                beta11[k] = beta11[k] + x * pl
                plm, pl = pl, ((2 * k + 1.) * xr * pl - k * plm) / (k + 1.)
        beta11 = (2 * kk + 1) * beta11 * .5
        coef = 2 * (1 - beta11[0]) if itronc else 0.0
        if itronc and coef < SEUIL_TRONCA:
            itronc = 0                    # truncation too small to matter: start again without it (SOS_AEROSOLS.F:4195-4214)
            continue
        break
    gamma12, beta22, delta33 = np.zeros(os_nb + 1), np.zeros(os_nb + 1), np.zeros(os_nb + 1)
    for j in range(-n, n + 1):
        if j == 0:
            continue
        xr = xmu[J(j)]
        pol = np.zeros(os_nb + 2)
        pol[2] = 3. * (1. - xr ** 2) / 2. / math.sqrt(6.0)
        xxx = xhr[J(j)] * p12[J(j)] * p11[J(j)] / ttt[J(j)]
        for k in range(2, os_nb + 1):
            d = (2. * k + 1.) / math.sqrt(1. * (k + 3.) * (k - 1.))
            e = math.sqrt(1. * (k + 2.) * (k - 2.)) / (2. * k + 1.)
            pol[k + 1] = d * (xr * pol[k] - e * pol[k - 1])
            gamma12[k] = gamma12[k] + xxx * pol[k]
        x = xhr[J(j)] * p22[J(j)] * (p11[J(j)] / ttt[J(j)])
        xx = xhr[J(j)] * p33[J(j)] * p11[J(j)] / ttt[J(j)]
        plm, pl = 0., 1.
        for k in range(os_nb + 1):
            beta22[k] = beta22[k] + x * pl
            delta33[k] = delta33[k] + xx * pl
            plm, pl = pl, ((2. * k + 1.) * xr * pl - k * plm) / (k + 1.)
    beta22 = beta22 * (2. * kk + 1.) * .5
    delta33 = delta33 * (2. * kk + 1.) * .5
    gamma12 = gamma12 * (2. * kk + 1.) * .5
    alp, zeta = np.zeros(os_nb + 1), np.zeros(os_nb + 1)
    f = np.float32
    for i in range(2, os_nb + 1):               # CO1, CO2, X2 are REAL*4 expressions (as in SOS_MAT_FRESNEL)
        co1 = float(f(4) * (f(2 * i) + f(1.)) / f(i) / (f(i) - f(1.)) / (f(i) + f(1.)) / (f(i) + f(2.)))
        co2 = float(f(i) * (f(i) - f(1.)) / ((f(i) + f(1.)) * (f(i) + f(2.))))
        co3 = co2 * delta33[i]
        co2 = co2 * beta22[i]
        nn, mm = int(i * .5), int((i - 1) * .5)
        s1 = s2 = s3 = s4 = 0.
        for j in range(1, nn + 1):
            x2 = float((f(i) - f(1.)) * (f(i) - f(1.)) - f(3.) * (f(2 * j) - f(1.)) * f(i - j))
            s1 = s1 + x2 * beta22[i - 2 * j]
            s2 = s2 + x2 * delta33[i - 2 * j]
        for j in range(0, mm + 1):
            x2 = float((f(i) - f(1.)) * (f(i) - f(1.)) - f(3.) * f(j) * (f(2 * i - 2 * j) - f(1.)))
            s3 = s3 + x2 * beta22[i - 2 * j - 1]
            s4 = s4 + x2 * delta33[i - 2 * j - 1]
        zeta[i] = co3 - co1 * (s2 - s3)
        alp[i] = co2 - co1 * (s1 - s4)
    z1 = beta11[0]
    return dict(alpha=alp / z1, beta=beta11 / z1, gamma=gamma12 / z1, zeta=zeta / z1, beta22=beta22 / z1, delta33=delta33 / z1,
                coef_tronca=float(coef), itronc=itronc)


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


def aerosols(p, wa, ta, nb_gauss_mie, os_nb, *, at_waref=False, device=0):
    """SOS_AEROSOLS for the wavelength wa.  p: the sos_proc keyword dictionary (run_sos.SOS_PROC_KWARGS names); at_waref
    selects the refractive indices of the reference wavelength (the first of the two calls SOS_PROC makes when
    WA_SIMU != WAREF, SOS_PROC.F:2893 / :3028).  Returns the content of Aerosols.txt:
    dict(alpha, beta, gamma, zeta [os_nb+1], a_tronc, piz, piztr (as printed: F9.5), kmat1, kmat2, coef_tronca (full precision))."""
    imod = int(p["imod_aer"])
    itronc = int(p["itronc_aer"])
    xmu, xhr = mie_angles(nb_gauss_mie)
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
        rec = mie_records(xmu, rn, in_, MIE_ALPHAMIN, af, device)
        kmat1, kmat2, _, p11, p12, p33 = granu(rec, igranu, v1, v2, v3, wa)
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
                rec = mie_records(xmu, rnr, inr, MIE_ALPHAMIN, _alphaf(_rmax_lnd(md["r"], md["v"]), waref), device)
                kref.append(granu(rec, 1, md["r"], md["v"], -999.0, waref)[0])
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
            rec = mie_records(xmu, md["rn"], md["in_"], MIE_ALPHAMIN, _alphaf(_rmax_lnd(md["r"], md["v"]), wa), device)
            k1, k2, _, a11, a12, a33 = granu(rec, 1, md["r"], md["v"], -999.0, wa)
            kmat1 = kmat1 + c * k1
            kmat2 = kmat2 + c * k2
            p11 = p11 + c * a11 * k2
            p12 = p12 + c * a12 * k2
            p33 = p33 + c * a33 * k2
        p11, p12, p33 = p11 / kmat2, p12 / kmat2, p33 / kmat2
    else:
        raise NotImplementedError("-AER.Model %d (WMO, Shettle & Fenn, external data, user mixtures) is not built; "
                                  "give the phase-matrix expansion through -AER.UserFile" % imod)
    d = decompo_legendre(itronc, xmu, xhr, os_nb, p11, p12, p11.copy(), p33)
    piz = kmat2 / kmat1
    ct = d["coef_tronca"]
    piztr = piz * (1. - ct / 2.) / (1. - piz * ct / 2.)
    # what SOS_PREPA_OS reads back from Aerosols.txt: E15.8 coefficients, F9.5 truncation coefficient and albedo
    q8 = lambda a: _round_sig(a, 8)
    a_f, piztr_f = float("%9.5f" % ct), float("%9.5f" % piztr)
    return dict(alpha=q8(d["alpha"]), beta=q8(d["beta"]), gamma=q8(d["gamma"]), zeta=q8(d["zeta"]), a_tronc=a_f, piztr=piztr_f,
                piz=piztr_f / (1 + 0.5 * a_f * (piztr_f - 1)), kmat1=kmat1, kmat2=kmat2, coef_tronca=ct)
