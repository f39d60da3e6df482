"""Host-side mirror of the reference's Python front end `binding/run_sos.py`.

Same `-Group.Key` parameter dictionary (run_sos.py:459-559), same user-override merge (:606-609, unknown keys
dropped), same `set_sos_params` ordering (:319-441), same `sos_proc(**kwargs)` keyword names (:640-695) and
the same 23-tuple of outputs with the reference shapes (`PHI_FIN(0:360)`, `THETA_FIN(0:80)`, `(361,81)` tables,
SOS_PROC.F:1168-1204).  Behind that surface the hot path runs on the GPU through the C ABI:

    angles (host)  ->  [aerosol expansion: given]  ->  profile (host)  ->  SOS.F rescale (host)
      -> sosgpu_glitter (ISURF=1) -> sosgpu_noyaux -> sosgpu_os_solve -> sosgpu_aggregate -> sosgpu_trphi

Scope.  Supported: gas absorption by the CKD method (`-AP.AbsProfile.Type` 0..6, both
`-SOS.AbsModeCKD` modes, all bins of a band solved in one batch) or none (7); aerosols given by the reference's own
`-AER.UserFile` (an Aerosols.txt) or by the extension keyword `aer_phase`, or none; exponential profiles
(`-AP.AerProfile.Type 1`); surfaces `-SURF.Type` 0 (Lambert), 1 (+ Cox-Munk glitter), 2 (+ flat sea), 3 (Roujean),
4 / 5 / 7 (Roujean + Rondeaux-Herman / Breon / Maignan), or reflection matrices from a user file (`-SURF.File`, the format
SOS_SURFACE writes); `-SOS.Trans`, `-SOS.Flux`, `SOS_Result.bin` files.
Every aerosol model of `-AER.Model`: 0 (mono-modal log-normal / Junge), 1 (WMO), 2 (Shettle & Fenn), 3 (bimodal
log-normal), 4 (external phase functions), 5 (user mixture) -- Mie theory on the GPU (aerosols.py, csrc/mie.hip).
Both aerosol profiles: exponential (`-AP.AerProfile.Type 1`) and a layer between two altitudes (2; the reference reads an
unassigned Hmol(0) there -- see profile_layer).  Not built: `-SURF.Type 6` (Nadal), which the reference's SOS_PROC refuses
as well.

Differences from the reference script that are deliberate: errors raise exceptions instead of being
lost in an `intent(in)` `ier`; `gen_sos_output` imports `ceil` and uses `and` (the reference has `1 & updown == 2`);
files are written only when `-SOS_Main.ResRoot` is given (RESROOT/SOS/SOS_Result.bin, the -SOS.Trans / -SOS.Flux files).
"""
import collections
import functools
import math
import os
import threading

import numpy as np

from .synth import MDF_DEFAULT, _round_sig, rescale_profile

SOS_NOT_DEFINED_VALUE_INT = -999
SOS_NOT_DEFINED_VALUE_DBLE = -999.0
SOS_DEFAULT_AER_JUNGE_RMAX = 50.
SOS_DEFAULT_FICANGRESLUM = "SOS_UsedAngles.txt"
SOS_DEFAULT_FICANGRESMIE = "Aer_UsedAngles.txt"
SOS_DEFAULT_FICGRANU = "Aerosols.txt"
SOS_DEFAULT_RESBIN = "SOS_Result.bin"
SOS_DEFAULT_RESUP = "SOS_Up.txt"
SOS_DEFAULT_RESDOWN = "SOS_Down.txt"

# inc/SOS.h constants used by the host-side restatements (REAL*4 literals are widened exactly as Fortran does)
_F = lambda x: float(np.float32(x))
CTE_OS_NBMU_MAX = 80
CTE_LENFIC2 = 500                    # SOS.h:59
CTE_OS_NT = 600                      # SOS.h:202
CTE_OS_NT_MIN = 100                  # SOS.h:229
CTE_TCOUCHE = _F(0.005)              # SOS.h:208
CTE_TOA_FIRST_LAYER = _F(0.0002)     # SOS.h:213
CTE_DELTA_Z = _F(0.05)               # SOS.h:218
CTE_TOA_ALT = 120.0                  # SOS.h:197
CTE_HT_STD_PSURF = 1013.0            # SOS.h:192
CTE_DEFAULT_IGMAX = 100              # SOS.h:383
CTE_DEFAULT_NBMU_LUM, CTE_DEFAULT_NBMU_MIE = 24, 40          # SOS.h:514,508
CTE_DEFAULT_OS_NB, CTE_DEFAULT_OS_NS, CTE_DEFAULT_OS_NM = 80, 48, 128   # SOS.h:521-535

# (dictionary key, sos_proc keyword, default) in the positional order of SOS_PROC (SOS_PROC.F:415-470)
_I, _D = SOS_NOT_DEFINED_VALUE_INT, SOS_NOT_DEFINED_VALUE_DBLE
PARAMS = [
    ("-SOS_Main.ResRoot", "resroot", ""), ("-SOS_Main.Log", "ficmain_log", "SOS_Main.Log"),
    ("-SOS_Main.Wa", "wa_simu", _D), ("-ANG.Rad.NbGauss", "nbmu_gauss_lum", _I),
    ("-ANG.Rad.UserAngFile", "ficangles_user_lum", "NO_USER_ANGLES"), ("-ANG.Thetas", "tetas", _D),
    ("-ANG.Rad.ResFile", "ficangles_res_lum", SOS_DEFAULT_FICANGRESLUM), ("-ANG.Aer.NbGauss", "nbmu_gauss_mie", _I),
    ("-ANG.Aer.UserAngFile", "ficangles_user_mie", "NO_USER_ANGLES"),
    ("-ANG.Aer.ResFile", "ficangles_res_mie", SOS_DEFAULT_FICANGRESMIE), ("-ANG.Log", "ficanglog", "Angles.Log"),
    ("-AER.Waref", "waref_aot", _D), ("-AER.AOTref", "aot_ref", _D), ("-AER.Tronca", "itronc_aer", 1),
    ("-AER.Log", "ficgranu_log", "NO_LOG_FILE"), ("-AER.MieLog", "ficmie_log", "NO_LOG_FILE"),
    ("-AER.DirMie", "dir_mie", ""), ("-AER.ResFile", "ficgranu", SOS_DEFAULT_FICGRANU), ("-AER.Model", "imod_aer", _I),
    ("-AER.MMD.MRwa", "rn_wa", _D), ("-AER.MMD.MIwa", "in_wa", _D), ("-AER.MMD.MRwaref", "rn_waref", _D),
    ("-AER.MMD.MIwaref", "in_waref", _D), ("-AER.MMD.SDtype", "igranu", _I),
    ("-AER.MMD.LNDradius", "lnd_radius_mmd_aer", _D), ("-AER.MMD.LNDvar", "lnd_lnvar_mmd_aer", _D),
    ("-AER.MMD.JD.slope", "jd_slope_mmd_aer", _D), ("-AER.MMD.JD.rmin", "jd_rmin_mmd_aer", _D),
    ("-AER.MMD.JD.rmax", "jd_rmax_mmd_aer", SOS_DEFAULT_AER_JUNGE_RMAX), ("-AER.WMO.Model", "imodele_wmo", _I),
    ("-AER.WMO.DL", "c_wmo_dl", _D), ("-AER.WMO.WS", "c_wmo_ws", _D), ("-AER.WMO.OC", "c_wmo_oc", _D),
    ("-AER.WMO.SO", "c_wmo_so", _D), ("-AER.SF.Model", "imodele_sf", _I), ("-AER.SF.RH", "rh", _D),
    ("-AER.BMD.VCdef", "mode_param_bilnd", _I), ("-AER.BMD.CoarseVC", "user_cv_coarse", _D),
    ("-AER.BMD.FineVC", "user_cv_fine", _D), ("-AER.BMD.RAOT", "rtauct_waref", _D),
    ("-AER.BMD.CM.MRwa", "bmd_cm_mrwa", _D), ("-AER.BMD.CM.MIwa", "bmd_cm_miwa", _D),
    ("-AER.BMD.CM.MRwaref", "bmd_cm_mrwaref", _D), ("-AER.BMD.CM.MIwaref", "bmd_cm_miwaref", _D),
    ("-AER.BMD.CM.SDradius", "bmd_cm_rmodal", _D), ("-AER.BMD.CM.SDvar", "bmd_cm_var", _D),
    ("-AER.BMD.FM.MRwa", "bmd_fm_mrwa", _D), ("-AER.BMD.FM.MIwa", "bmd_fm_miwa", _D),
    ("-AER.BMD.FM.MRwaref", "bmd_fm_mrwaref", _D), ("-AER.BMD.FM.MIwaref", "bmd_fm_miwaref", _D),
    ("-AER.BMD.FM.SDradius", "bmd_fm_rmodal", _D), ("-AER.BMD.FM.SDvar", "bmd_fm_var", _D),
    ("-AER.ExtData", "ficextdata_aer", "NO_USER_AEROSOLS_PHAZE_FCT"),
    ("-AER.DefMixture", "ficmixture_aer", "NO_USER_AEROSOLS_MIXTURE"), ("-AER.UserFile", "ficuser_aer", "NO_USER_AEROSOLS"),
    ("-AP.Log", "ficprofil_log", "Aerosols.Log"), ("-AP.MOT", "tr", _D), ("-AP.HR", "hr", _D), ("-AP.AerHS.HA", "ha", _D),
    ("-AP.AerProfile.Type", "iprofil", 1), ("-AP.AerLayer.Zmin", "zmin", _D), ("-AP.AerLayer.Zmax", "zmax", _D),
    ("-AP.Psurf", "psurf", _D), ("-AP.H2O", "h2o", _D), ("-AP.O3", "o3", _D), ("-AP.CO2", "co2", _D), ("-AP.CH4", "ch4", _D),
    ("-AP.AbsProfile.Type", "absprofil", _I), ("-AP.AbsProfile.UserFile", "ficabsprofil", "NO_USER_ABS_PROFILE_FILE"),
    ("-AP.SpectralResol", "nustep", _I), ("-SURF.Type", "isurf", 0), ("-SURF.Dir", "dir_surf", ""),
    ("-SURF.Log", "ficsurf_log", "NO_LOG_FILE"), ("-SURF.Ind", "surf_ind", _D), ("-SURF.Glitter.Wind", "wind", _D),
    ("-SURF.Roujean.K0", "k0_roujean", _D), ("-SURF.Roujean.K1", "k1_roujean", _D), ("-SURF.Roujean.K2", "k2_roujean", _D),
    ("-SURF.Nadal.Alpha", "alpha_nadal", _D), ("-SURF.Nadal.Beta", "beta_nadal", _D),
    ("-SURF.Maignan.C", "coef_c_maignan", _D), ("-SURF.Alb", "rho", _D), ("-SURF.File", "ficsurf", "DEFAULT"),
    ("-SOS.Log", "ficsos_log", "SOS.Log"), ("-SOS.ResBin", "ficsos_res_bin", SOS_DEFAULT_RESBIN),
    ("-SOS.Trans", "fictrans", "NO_OUTPUT"), ("-SOS.Flux", "ficflux", "FicFlux.txt"), ("-SOS.OutputAlt", "zout", -1.0),
    ("-SOS.IGmax", "igmax", 200), ("-SOS.Ipolar", "ipolar", 1), ("-SOS.View", "itrphi", 2),
    ("-SOS.View.Phi", "phios", _D), ("-SOS.View.Dphi", "pas_phi", _I), ("-SOS.AbsModeCKD", "imode_ckd_calcul", 1),
]
# keys the reference dictionary holds but never forwards to sos_proc (run_sos.py:548-549,557)
EXTRA_KEYS = {"-SOS.ResFileUp": SOS_DEFAULT_RESUP, "-SOS.ResFileDown": SOS_DEFAULT_RESDOWN, "-SOS.MDF": 0.0279}
SOS_PROC_KWARGS = [p[1] for p in PARAMS] + ["ier", "trace"]
OUTPUT_NAMES = ["nblum", "ind_angout", "phi", "vza", "sca_ang_up", "i_up", "q_up", "u_up", "pol_ang_up", "pol_rate_up",
                "l_pol_up", "sca_ang_down", "i_down", "q_down", "u_down", "pol_ang_down", "pol_rate_down", "l_pol_down",
                "flux_dir_down", "flux_diff_down", "flux_tot_down", "flux_diff_up", "coef_tronca"]


class SosProcError(RuntimeError):
    """Raised where the reference prints a message and returns IER=1 (SOS_PROC.F:3895-4896)."""

    def __init__(self, msg, ier=1):
        super().__init__(msg)
        self.ier = ier


def default_parameters():
    """The reference's `dict_sos_all_parameters` (run_sos.py:459-559)."""
    d = {k: v for k, _, v in PARAMS}
    d.update(EXTRA_KEYS)
    return d


def update_parameters(defaults, user):
    """run_sos.py:606-609: user values override defaults; keys the default dictionary does not hold are dropped."""
    out = dict(defaults)
    for k, v in user.items():
        if k in defaults:
            out[k] = v
    return out


def set_sos_params(dict_sos, trace=True):
    """run_sos.py:319-441: the positional tuple handed to sos_proc (96 values: 94 parameters, ier, trace)."""
    return tuple(dict_sos[k] for k, _, _ in PARAMS) + (0, trace)


def sos_proc_kwargs(dict_sos, trace=True):
    return dict(zip(SOS_PROC_KWARGS, set_sos_params(dict_sos, trace)))


# ---------------------------------------------------------------------------------------------------------
# file formats either side of the hot path
# ---------------------------------------------------------------------------------------------------------
def fortran_e(x, width, digits):
    """Fortran Ew.d edit descriptor: 0.ddddddddE+ee, right-justified in `width` columns.  The digits come from the correctly
    rounded decimal conversion of the binary value (C's %e, as the Fortran runtime's), shifted to the 0.d form."""
    if x == 0.0:
        body = ("-" if math.copysign(1.0, x) < 0 else "") + "0." + "0" * digits + "E+00"
    else:
        m, e = ("%.*e" % (digits - 1, abs(x))).split("e")
        body = ("-" if x < 0 else "") + "0." + m.replace(".", "") + "E%+03d" % (int(e) + 1)
    return body.rjust(width)


def fortran_d(x, width, digits):
    return fortran_e(x, width, digits).replace("E", "D")


def read_aerosols_file(path, os_nb):
    """Parse an `Aerosols.txt` (written by SOS_AEROSOLS.F:2864-2890, read by SOS_PREPA_OS.F:666-700): truncation
    coefficient A, truncated single-scattering albedo PIZTR (both after the ':' of their header line), three title
    lines, then ALPHA, BETA, GAMMA, ZETA for K = 0..OS_NB (list-directed read).  PIZ = PIZTR/(1 + A (PIZTR-1)/2)
    (SOS_PREPA_OS.F:699).  Returns the `aer_phase` dict sos_proc takes."""
    with open(path) as f:
        lines = f.read().splitlines()
    if len(lines) < 8 + os_nb + 1:
        raise SosProcError("aerosol file %s: %d coefficient rows expected" % (path, os_nb + 1))
    a = float(lines[3].split(":", 1)[1].replace("D", "E"))
    piztr = float(lines[4].split(":", 1)[1].replace("D", "E"))
    rows = np.array([[float(v.replace("D", "E")) for v in lines[8 + k].split()[:4]] for k in range(os_nb + 1)])
    return dict(alpha=rows[:, 0].copy(), beta=rows[:, 1].copy(), gamma=rows[:, 2].copy(), zeta=rows[:, 3].copy(),
                a_tronc=a, piztr=piztr, piz=piztr / (1 + 0.5 * a * (piztr - 1)))


def write_aerosols_file(path, aer_phase, kmat1=0.0, kmat2=0.0):
    """`Aerosols.txt` in the layout of SOS_AEROSOLS.F:2864-2890 (formats 40-50, :3049-3056)."""
    al, be, ga, ze = (np.asarray(aer_phase[k], dtype=np.float64) for k in ("alpha", "beta", "gamma", "zeta"))
    a, piztr = float(aer_phase.get("a_tronc", 0.0)), float(aer_phase["piztr"])
    with open(path, "w") as f:
        f.write("EXTINCTION CROSS SECTION (mic^2)     :" + fortran_e(kmat1, 13, 5) + "\n")
        f.write("SCATTERING CROSS SECTION (mic^2)     :" + fortran_e(kmat2, 13, 5) + "\n")
        f.write("ASYMMETRY FACTOR (no truncation)     :" + fortran_e(a / 2. + (1. - a / 2.) * be[1] / 3., 13, 5) + "\n")
        f.write("TRUNCATION COEFFICIENT               :%9.5f\n" % a)
        f.write("SINGLE SCATTERING ALBEDO (truncation):%9.5f\n" % piztr)
        f.write("-" * 33 + "\n")
        f.write("PHASE MATRIX COEFFICIENTS FOR K=0 TO%4d\n" % (len(be) - 1))
        f.write("ALPHA(K)        BETA11(K)       GAMMA12(K)      ZETA(K)\n")
        for k in range(len(be)):
            f.write(fortran_e(al[k], 15, 8) + "".join(" " + fortran_e(v[k], 15, 8) for v in (be, ga, ze)) + "\n")


def write_used_angles(path, mu, ga, n0, ind_ang, nb_gauss, tetas, os_nb, os_ns, os_nm, user_file="NO_USER_ANGLES"):
    """`SOS_UsedAngles.txt` (SOS_ANGLES.F:494-506 header, :639-647 rows `I4,1X,2D21.14,1X,I4`)."""
    with open(path, "w") as f:
        f.write("NB_TOTAL_ANGLES :%4d\n" % len(mu))
        f.write("NB_GAUSS_ANGLES :%4d\n" % nb_gauss)
        f.write("ANGLES_USERFILE :" + user_file.ljust(CTE_LENFIC2) + "\n")
        f.write("SOLAR ZENITH ANGLE :%7.3f\n" % tetas)
        f.write("INTERNAL_IMUS :%4d\n" % n0)
        f.write("INTERNAL_OS_NB :%4d\n" % os_nb)
        f.write("INTERNAL_OS_NS :%4d\n" % os_ns)
        f.write("INTERNAL_OS_NM :%4d\n" % os_nm)
        f.write("INDEX   COS_ANGLE            WEIGHT             USER_ANGLE\n")
        for j in range(len(mu)):
            f.write("%4d " % (j + 1) + fortran_d(mu[j], 21, 14) + fortran_d(ga[j], 21, 14) + " %4d\n" % ind_ang[j])


def sos_gauss(nb_gauss):
    """(cached per order: the Newton iterations cost 2 ms of every call otherwise; fresh arrays are returned)"""
    mu, wt = _sos_gauss_nodes(int(nb_gauss))
    return np.array(mu), np.array(wt)


@functools.lru_cache(maxsize=32)
def _sos_gauss_nodes(nb_gauss):
    """SOS_GAUSS (SOS_ANGLES.F:1022-1103) for MM = nb_gauss + 1: the positive nodes and weights of the 2 nb_gauss-point
    Gauss-Legendre rule, ascending in the cosine -- the reference's own Newton iteration (asymptotic start, three-term
    recurrence, stop at |dx| <= 1e-15), statement for statement, so that the D21.14 digits of the angle files agree in
    their last place too (numpy's leggauss differs from it by an ulp here and there)."""
    n = 2 * nb_gauss
    pi = math.pi
    aa = 2.0 / pi ** 2
    ab = -62.0 / (3.0 * pi ** 4)
    ac = 15116.0 / (15.0 * pi ** 6)
    ad = -12554474.0 / (105.0 * pi ** 8)
    en = float(n)
    u = 1.0 - (2.0 / pi) ** 2
    d = 1.0 / math.sqrt((en + 0.5) ** 2 + u / 4.0)
    r, w = [], []
    for k in range(1, n + 1):
        az = 4.0 * k - 1.0
        z = 0.25 * pi * (az + aa / az + ab / az ** 3 + ac / az ** 5 + ad / az ** 7)
        x = math.cos(z * d)
        while True:
            pm2, pm1 = 1.0, x                              # PA(1), PA(2)
            for nn in range(3, n + 2):
                enn = nn - 1.0
                pm2, pm1 = pm1, ((2.0 * enn - 1.0) * x * pm1 - (enn - 1.0) * pm2) / enn
            pnp = en * (pm2 - x * pm1) / (1.0 - x * x)     # PA(N), PA(NP1)
            xi = x - pm1 / pnp
            if abs(xi - x) - 1.0e-15 <= 0.0:
                r.append(x)
                w.append(2.0 * (1.0 - x * x) / (en * pm2) ** 2)
                break
            x = xi
    # R(I), I = 1..MM-1, descend from the largest node: AMU(K = MM - I) -> ascending cosines for K = 1..nb_gauss
    mu = tuple([r[i] for i in range(nb_gauss)][::-1])
    wt = tuple([w[i] for i in range(nb_gauss)][::-1])
    return mu, wt


def write_mie_angles(path, nb_gauss, os_nb, user_file="NO_USER_ANGLES"):
    """`Aer_UsedAngles.txt` (SOS_ANGLES.F:367-376, rows `I4,1X,2D21.14`): the Gauss nodes of the phase-function angle set plus
    the user's angles (weight 0), cosines ascending."""
    x, w = sos_gauss(nb_gauss)
    mu, wt = list(x), list(w)
    if user_file != "NO_USER_ANGLES":
        with open(user_file) as f:
            for line in f:
                if line.strip():
                    val = float(line.split()[0])
                    if val < 0. or val > 90.:
                        raise SosProcError("user angle out of [0,90] in %s" % user_file)
                    mu.append(math.cos(val * math.pi / 180.))
                    wt.append(0.0)
    order = np.argsort(np.asarray(mu), kind="stable")
    with open(path, "w") as f:
        f.write("NB_TOTAL_ANGLES :%4d\n" % len(mu))
        f.write("NB_GAUSS_ANGLES :%4d\n" % nb_gauss)
        f.write("ANGLES_USERFILE :" + user_file.ljust(CTE_LENFIC2) + "\n")
        f.write("INTERNAL_OS_NB :%4d\n" % os_nb)
        f.write("INDEX   COS_ANGLE            WEIGHT\n")
        for i, j in enumerate(order):
            f.write("%4d " % (i + 1) + fortran_d(mu[j], 21, 14) + fortran_d(wt[j], 21, 14) + "\n")


# ---------------------------------------------------------------------------------------------------------
# host-side restatements of the steps before the hot path (inputs of SOS_OS)
# ---------------------------------------------------------------------------------------------------------
_ANGLES_CACHE = collections.OrderedDict()
_ANGLES_LOCK = threading.Lock()


def angles(nbmu_gauss, tetas, user_file="NO_USER_ANGLES"):
    """SOS_ANGLES for the radiance angles (SOS_ANGLES.F:380-466, 713-866): Gauss nodes/weights of the
    2*NbGauss-point rule on [-1,1] (positive half), optional user angles (weight 0), solar angle inserted
    with weight 0 unless it coincides with a node, mu descending; values as re-read from SOS_UsedAngles.txt
    (D21.14).  Returns mu[N], ga[N], n0 (1-based), ind_angout[N] (1 = user angle).
    The last few angle sets are kept (the calls of a spectrum share theirs; a user file is keyed by its size and time stamp)."""
    key = (int(nbmu_gauss), float(tetas), str(user_file))
    if user_file != "NO_USER_ANGLES":
        try:
            st = os.stat(user_file)
            key += (st.st_size, st.st_mtime_ns)
        except OSError:
            key = None
    with _ANGLES_LOCK:
        hit = _ANGLES_CACHE.get(key) if key is not None else None
    if hit is None:
        hit = _angles(nbmu_gauss, tetas, user_file)
        if key is not None:
            with _ANGLES_LOCK:
                _ANGLES_CACHE[key] = hit
                while len(_ANGLES_CACHE) > 16:
                    _ANGLES_CACHE.popitem(last=False)
    mu, ga, n0, ind = hit
    return mu.copy(), ga.copy(), n0, ind.copy()


def _angles(nbmu_gauss, tetas, user_file):
    x, w = sos_gauss(nbmu_gauss)
    mu = list(x)
    wt = list(w)
    if user_file != "NO_USER_ANGLES":
        with open(user_file) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                val = float(line.split()[0])
                if val < 0. or val > 90.:
                    raise SosProcError("user angle out of [0,90] in %s" % user_file)
                mu.append(math.cos(val * math.pi / 180.))
                wt.append(0.0)
    order = np.argsort(-np.asarray(mu), kind="stable")
    mu = np.asarray(mu)[order]
    wt = np.asarray(wt)[order]
    ind = (wt == 0.0).astype(np.int32)
    xmus = math.cos(tetas * math.pi / 180.)
    imus = -1
    for j in range(len(mu)):
        if abs(xmus - mu[j]) < _F(0.00001):       # CTE_SEUIL_ECART_MUS, SOS.h:561
            imus = j + 1
    if imus == -1:
        pos = int(np.sum(mu > xmus))
        mu = np.insert(mu, pos, xmus)
        wt = np.insert(wt, pos, 0.0)
        ind = np.insert(ind, pos, 0)
        imus = pos + 1
    if len(mu) > CTE_OS_NBMU_MAX:
        raise SosProcError("number of radiance angles > CTE_OS_NBMU_MAX")
    return _round_sig(mu, 14), _round_sig(wt, 14), imus, ind


def rayleigh_optical_thickness(wa_simu, psurf):
    """SOS_PROC.F:3333-3334 (CNES formulation); 84.35, 1.225, 1.4 are REAL*4 literals."""
    return (psurf / CTE_HT_STD_PSURF) * 1.e-4 * (_F(84.35) / wa_simu ** 4 - _F(1.225) / wa_simu ** 5 + _F(1.4) / wa_simu ** 6)


def _disc(dt, ta, ha, tr, hr, tim1, zmax_init):
    """SOS_DISC without gas (SOS_PROFIL.F:1260-1325): altitude where tau(z) = tim1 + dt, by bisection."""
    ti = tim1 + dt
    zmax, zmin = zmax_init, 0.0
    while True:
        zmoy = (zmax + zmin) / 2.
        tz = ta * math.exp(-zmoy / ha) + tr * math.exp(-zmoy / hr)
        xd = abs(ti - tz)
        if xd < _F(.000001) or zmoy == 0.0:
            return zmoy
        if (ti - tz) < 0.0:
            zmin = zmoy
        else:
            zmax = zmoy


def profile_nogas(tr, hr, ta, ha):
    """SOS_PROFILE for IPROFIL=1 without gas absorption (SOS_PROFIL.F:349-508), then the PROFIL_TMP text
    round trip (format 20 `2X,I5,F10.5,3(E15.8)`, SOS_PROFIL.F:1084,1150).  Returns h, xdel, ydel, zprof."""
    ttot = tr + ta
    if (ttot / CTE_OS_NT_MIN) <= CTE_TOA_FIRST_LAYER:
        nt = CTE_OS_NT_MIN
        t_layer = ttot / nt
        t_first = t_layer
    elif (ttot / CTE_OS_NT_MIN) < CTE_TCOUCHE:
        nt = CTE_OS_NT_MIN + 1
        t_first = CTE_TOA_FIRST_LAYER
        t_layer = (ttot - t_first) / CTE_OS_NT_MIN
    else:
        t_first = CTE_TOA_FIRST_LAYER
        nt = int((ttot - t_first) / CTE_TCOUCHE)
        t_layer = (ttot - t_first) / nt
        nt = nt + 1
    if nt > CTE_OS_NT:
        raise SosProcError("profile needs more than CTE_OS_NT levels")
    hmol = np.zeros(nt + 1); haer = np.zeros(nt + 1); z = np.zeros(nt + 1)
    pcmol = np.zeros(nt + 1); pcaer = np.zeros(nt + 1)
    if ta == 0.0:                                   # SOS_PROFIL.F:412-429
        hmol[1] = t_first
        for i in range(2, nt + 1):
            hmol[i] = (i - 1) * t_layer + t_first
        pcmol[:] = 1.0
        z[0] = CTE_TOA_ALT
        for i in range(1, nt + 1):
            z[i] = hr * math.log(tr / hmol[i])
    else:                                           # SOS_PROFIL.F:437-489
        z[0] = CTE_TOA_ALT
        dtau, zz = 0., CTE_TOA_ALT
        while dtau < t_first:
            zz = zz - CTE_DELTA_Z
            dtau = tr * math.exp(-zz / hr) + ta * math.exp(-zz / ha)
        z[1] = zz
        vr, va = tr * math.exp(-zz / hr), ta * math.exp(-zz / ha)
        hmol[1], haer[1] = vr, va
        pcmol[1], pcaer[1] = vr / dtau, va / dtau
        pcmol[0], pcaer[0] = pcmol[1], pcaer[1]
        hprev = dtau
        for i in range(2, nt):
            zz = _disc(t_layer, ta, ha, tr, hr, hprev, z[1])
            z[i] = zz
            vr, va = tr * math.exp(-zz / hr), ta * math.exp(-zz / ha)
            hmol[i], haer[i] = vr, va
            hprev = vr + va
            vr, va = vr - hmol[i - 1], va - haer[i - 1]
            pcmol[i], pcaer[i] = vr / (vr + va), va / (vr + va)
        z[nt] = 0.0
        hmol[nt], haer[nt] = tr, ta
        vr, va = tr - hmol[nt - 1], ta - haer[nt - 1]
        pcmol[nt], pcaer[nt] = vr / (vr + va), va / (vr + va)
    h = hmol + haer
    return _round_sig(h, 8), _round_sig(pcaer, 8), _round_sig(pcmol, 8), np.round(z, 5)


def profile_layer(tr, hr, ta, zmin, zmax):
    """SOS_PROFILE for IPROFIL=2 (SOS_PROFIL.F:800-946): a homogeneous aerosol + molecule layer between zmin and zmax (km)
    inside a molecular atmosphere, no gas absorption, then the PROFIL_TMP text round trip.  Returns h, xdel, ydel, zprof.

    The reference starts its recurrence `Hmol(I)=Hmol(I-1)+VR_SC` (:873) from an Hmol(0) it assigns only at :926 -- a local
    array element read before it is written.  On a fresh stack it is 0, which is also the evident intent (the optical
    depth at the top of the atmosphere, assigned TR*exp(-120/HR) ~ 1e-7 TR afterwards); this restatement starts from 0 and
    is pinned against the reference's routine called on a fresh stack (tests/golden/profile_layer.npz)."""
    if zmin < 0.0 or zmax <= zmin:
        raise SosProcError("SOS_PROFIL: -AP.AerLayer.Zmin / Zmax must satisfy 0 <= Zmin < Zmax", ier=-1)
    ttot = tr + ta
    nt = int(ttot / CTE_TCOUCHE)
    if nt > CTE_OS_NT:
        nt = CTE_OS_NT
    dzt = _F(0.010)                                   # CTE_DZTRANSI (REAL*4 literal, SOS.h:240)
    vr_c1 = tr * math.exp(-(zmax + dzt) / hr)
    vr_c2 = tr * (math.exp(-zmin / hr) - math.exp(-(zmax + dzt) / hr))
    if zmin == 0.0:
        vr_c3, nb_tr = 0.0, 1
    else:
        vr_c3, nb_tr = tr * (1.0 - math.exp(-(zmin - dzt) / hr)), 2
    nbsc_c1 = max(3, int((nt - nb_tr) * vr_c1 / (tr + ta)))              # CTE_PROFIL_MIN_NBC = 3
    nbsc_c3 = 0 if zmin == 0.0 else max(3, int((nt - nb_tr) * vr_c3 / (tr + ta)))
    nbsc_c2 = (nt - nb_tr) - nbsc_c1 - nbsc_c3
    if nbsc_c2 <= 0 or (ta / nbsc_c2) < _F(0.00001):
        raise SosProcError("SOS_PROFIL: not enough sublayers in the aerosol layer (ERROR 1020)", ier=-1)
    hmol = np.zeros(nt + 1); haer = np.zeros(nt + 1)
    vr_sc = vr_c1 / nbsc_c1
    for i in range(1, nbsc_c1 + 1):
        hmol[i] = hmol[i - 1] + vr_sc
    i = nbsc_c1 + 1
    hmol[i] = tr * math.exp(-zmax / hr)
    vr_sc = hmol[i] - hmol[i - 1]
    haer[i] = haer[i - 1] + (ta * vr_sc / vr_c2)
    delta_z = (zmax - zmin) / nbsc_c2
    z = zmax
    for i in range(nbsc_c1 + 2, nbsc_c1 + nbsc_c2 + 2):
        z = z - delta_z
        hmol[i] = tr * math.exp(-z / hr)
        vr_sc = hmol[i] - hmol[i - 1]
        haer[i] = haer[i - 1] + (ta * vr_sc / vr_c2)
    if zmin != 0.0:
        i = nbsc_c1 + nbsc_c2 + 2
        hmol[i] = tr * math.exp(-(zmin - dzt) / hr)
        haer[i] = haer[i - 1]
        vr_sc = vr_c3 / nbsc_c3
        for i in range(nbsc_c1 + nbsc_c2 + 3, nt + 1):
            hmol[i] = vr_sc + hmol[i - 1]
            haer[i] = haer[i - 1]
    zprof = np.zeros(nt + 1); h = np.zeros(nt + 1); pcaer = np.zeros(nt + 1); pcmol = np.zeros(nt + 1)
    zprof[0] = CTE_TOA_ALT
    hmol[0] = tr * math.exp(-CTE_TOA_ALT / hr)
    h[0] = hmol[0] + haer[0]
    pcmol[0] = 1.0
    for i in range(1, nt + 1):
        h[i] = hmol[i] + haer[i]
        zprof[i] = hr * math.log(tr / hmol[i])
        if haer[i] == haer[i - 1]:
            pcaer[i], pcmol[i] = 0.0, 1.0
        else:
            pcmol[i] = 1 / (1 + (ta / vr_c2))
            pcaer[i] = 1 - pcmol[i]
    return _round_sig(h, 8), _round_sig(pcaer, 8), _round_sig(pcmol, 8), np.round(zprof, 5)


def read_surface_file(path, n, os_nb):
    """A surface reflection-matrix file in the reference's format (what SOS_SURFACE writes and SOS_OS reads, SOS_OS.F:916-925):
    one sequential unformatted record per Fourier order IS = 0..OS_NB holding the nine REAL*4 matrices P11 P12 P13 P21 ...
    P33, each ((R(I,J), I=1,N), J=1,N).  Returns float32 [os_nb+1][9][N][N] (the layout of SosContext(rsurf=...))."""
    want = 9 * n * n * 4
    out = np.zeros((os_nb + 1, 9, n, n), dtype=np.float32)
    with open(path, "rb") as f:
        for s in range(os_nb + 1):
            head = f.read(4)
            if len(head) < 4:
                raise SosProcError("surface file %s: %d records, %d expected (OS_NB + 1)" % (path, s, os_nb + 1))
            nbytes = int(np.frombuffer(head, "<i4")[0])
            if nbytes != want:
                raise SosProcError("surface file %s: record of %d bytes, 9 x %d x %d REAL*4 expected -- made for another "
                                   "angle set?" % (path, nbytes, n, n))
            out[s] = np.frombuffer(f.read(nbytes), "<f4").reshape(9, n, n)
            f.read(4)
    return out


def write_surface_file(path, rsurf):
    """Inverse of read_surface_file (rsurf: float32 [F][9][N][N], e.g. surface.glitter_matrices(...)["rsurf"].cpu())."""
    rsurf = np.ascontiguousarray(rsurf, dtype="<f4")
    with open(path, "wb") as f:
        for s in range(rsurf.shape[0]):
            payload = rsurf[s].tobytes()
            m = np.array([len(payload)], "<i4").tobytes()
            f.write(m + payload + m)


def _dist_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def _shard_range(nb, rank, world):
    from .dist import shard_range
    return shard_range(nb, rank, world)


def validate_parameters(p):
    """The user-parameter checks of SOS_PROC in the reference's order (SOS_PROC.F:1310-1335, 1540-2475): the first violated
    rule raises SosProcError carrying the reference's error number (`.code`, the label of its message block at
    SOS_PROC.F:4074-4745) and the keyword to fix.  Side effects of that block are mirrored too: with a single wavelength
    the reference-wavelength refractive indices default to the simulation ones (:1704-1707, :1812-1818)."""
    def fail(code, text):
        e = SosProcError("SOS_PROC : ERROR_%s on parameters -- %s" % (code, text))
        e.code = code
        raise e
    und = lambda k: p[k] == _D                       # CTE_NOT_DEFINED_VALUE_DBLE = -999. and _INT = -999 compare equal
    user_aer = str(p["ficuser_aer"]).strip() != "NO_USER_AEROSOLS"
    # (ERROR_1000, -SOS_Main.ResRoot undefined, is not mirrored: without a results directory this implementation simply
    #  writes no files -- a deliberate difference, see the module docstring)
    if und("aot_ref"):
        fail(2301, "-AER.AOTref (aerosol optical thickness at the reference wavelength) must be defined")
    if p["aot_ref"] > 0.0:
        if und("imod_aer") and not user_aer:
            fail(2304, "-AER.Model must be defined (0 mono-modal, 1 WMO, 2 Shettle & Fenn, 3 bimodal log-normal, "
                       "4 external phase functions, 5 user mixture)")
        if und("waref_aot"):
            fail(2302, "-AER.Waref (reference wavelength of the aerosol optical thickness) must be defined")
    if und("wa_simu"):
        fail(2100, "-SOS_Main.Wa (simulation wavelength, microns) must be defined")
    if p["wa_simu"] < _F(0.364) or p["wa_simu"] > _F(4.0):                          # CTE_WAMIN, CTE_WAMAX (SOS.h:70-71)
        fail(2101, "-SOS_Main.Wa = %r outside [0.364, 4.0] microns" % p["wa_simu"])
    if und("tetas"):
        fail(2200, "-ANG.Thetas (solar zenith angle, degrees) must be defined")
    if p["tetas"] < 0.0 or p["tetas"] >= 90.0:
        fail(2201, "-ANG.Thetas out of [0, 90[")
    if p["aot_ref"] > 0.0 and not user_aer:
        imod = int(p["imod_aer"])
        one_wl = p["wa_simu"] == p["waref_aot"]
        if imod < 0 or imod > 5:
            fail(2305, "-AER.Model out of 0..5")
        if int(p["itronc_aer"]) not in (0, 1):
            fail(23141, "-AER.Tronca must be 0 or 1")
        if imod == 0:
            if und("rn_wa") or und("in_wa"):
                fail(2309, "-AER.MMD.MRwa and -AER.MMD.MIwa must be defined")
            if p["in_wa"] > 0.0:
                fail(2310, "-AER.MMD.MIwa must be negative or null")
            if und("igranu"):
                fail(2311, "-AER.MMD.SDtype must be defined (1 log-normal, 2 Junge)")
            if int(p["igranu"]) not in (1, 2):
                fail(2312, "-AER.MMD.SDtype must be 1 or 2")
            if int(p["igranu"]) == 1 and (und("lnd_radius_mmd_aer") or und("lnd_lnvar_mmd_aer")):
                fail(23131, "-AER.MMD.LNDradius and -AER.MMD.LNDvar must be defined")
            if int(p["igranu"]) == 2 and (und("jd_slope_mmd_aer") or und("jd_rmin_mmd_aer")):
                fail(23132, "-AER.MMD.JD.slope and -AER.MMD.JD.rmin must be defined")
            if not one_wl:
                if und("rn_waref") or und("in_waref"):
                    fail(2314, "-AER.MMD.MRwaref and -AER.MMD.MIwaref must be defined when -SOS_Main.Wa differs from -AER.Waref")
            else:
                p["rn_waref"], p["in_waref"] = p["rn_wa"], p["in_wa"]
        if imod == 1:
            if und("imodele_wmo"):
                fail(2315, "-AER.WMO.Model must be defined")
            if not 1 <= int(p["imodele_wmo"]) <= 4:
                fail(2316, "-AER.WMO.Model must be in 1..4")
            if int(p["imodele_wmo"]) == 4 and any(und(k) for k in ("c_wmo_dl", "c_wmo_ws", "c_wmo_oc", "c_wmo_so")):
                fail(2317, "-AER.WMO.DL, .WS, .OC and .SO must be defined for the user WMO model")
        if imod == 2:
            if und("imodele_sf"):
                fail(2318, "-AER.SF.Model must be defined")
            if und("rh"):
                fail(2319, "-AER.SF.RH must be defined")
            if not 1 <= int(p["imodele_sf"]) <= 4:
                fail(2320, "-AER.SF.Model must be in 1..4")
            if p["rh"] < 0.0 or p["rh"] > 99.0:
                fail(2321, "-AER.SF.RH must be in [0, 99] %")
        if imod == 3:
            if und("mode_param_bilnd"):
                fail(2322, "-AER.BMD.VCdef must be defined")
            if int(p["mode_param_bilnd"]) not in (1, 2):
                fail(2323, "-AER.BMD.VCdef must be 1 or 2")
            if int(p["mode_param_bilnd"]) == 1:
                if und("user_cv_coarse"):
                    fail(2324, "-AER.BMD.CoarseVC must be defined")
                if und("user_cv_fine"):
                    fail(2325, "-AER.BMD.FineVC must be defined")
            if int(p["mode_param_bilnd"]) == 2 and und("rtauct_waref"):
                fail(2326, "-AER.BMD.RAOT must be defined")
            if any(und("bmd_cm_" + k) for k in ("mrwa", "miwa", "rmodal", "var")):
                fail(2327, "the coarse-mode parameters -AER.BMD.CM.{MRwa, MIwa, SDradius, SDvar} must be defined")
            if any(und("bmd_fm_" + k) for k in ("mrwa", "miwa", "rmodal", "var")):
                fail(2328, "the fine-mode parameters -AER.BMD.FM.{MRwa, MIwa, SDradius, SDvar} must be defined")
            if not one_wl:
                if any(und(k) for k in ("bmd_cm_mrwaref", "bmd_cm_miwaref", "bmd_fm_mrwaref", "bmd_fm_miwaref")):
                    fail(2329, "-AER.BMD.{CM,FM}.{MRwaref, MIwaref} must be defined when -SOS_Main.Wa differs from -AER.Waref")
            else:
                for m in ("cm", "fm"):
                    p["bmd_%s_mrwaref" % m], p["bmd_%s_miwaref" % m] = p["bmd_%s_mrwa" % m], p["bmd_%s_miwa" % m]
        if imod == 4:
            if str(p["ficextdata_aer"]).strip() == "NO_USER_AEROSOLS_PHAZE_FCT":
                fail(2330, "-AER.ExtData (file of external phase functions) must be defined")
            if not one_wl:
                fail(2331, "-AER.Model 4 requires -SOS_Main.Wa equal to -AER.Waref")
        if imod == 5 and str(p["ficmixture_aer"]).strip() == "NO_USER_AEROSOLS_MIXTURE":
            fail(2340, "-AER.DefMixture (file of the user mixture) must be defined")
    if p["aot_ref"] > 0.0 and user_aer:
        if p["wa_simu"] != p["waref_aot"]:
            fail(2350, "-AER.UserFile requires -SOS_Main.Wa equal to -AER.Waref")
        if str(p["ficgranu"]).strip() != SOS_DEFAULT_FICGRANU:
            fail(2351, "-AER.ResFile must keep its default value when -AER.UserFile is given")
    if und("rho"):
        fail(2401, "-SURF.Alb must be defined")
    if p["rho"] < 0.0:
        fail(2402, "-SURF.Alb must be positive or null")
    if und("isurf"):
        fail(2403, "-SURF.Type must be defined")
    isurf = int(p["isurf"])
    if isurf not in range(8):
        fail(2404, "-SURF.Type must be in 0..7")
    if und("surf_ind") and isurf in (1, 2, 4, 5, 6, 7):
        fail(2405, "-SURF.Ind (refractive index of the surface) must be defined for this surface type")
    if isurf == 1:
        if und("wind"):
            fail(2406, "-SURF.Glitter.Wind must be defined")
        if p["wind"] < 0.0:
            fail(24061, "-SURF.Glitter.Wind must be positive or null")
    if isurf >= 3 and any(und(k) for k in ("k0_roujean", "k1_roujean", "k2_roujean")):
        fail(2407, "-SURF.Roujean.K0, .K1 and .K2 must be defined")
    if isurf == 6:                                        # SOS_PROC.F:2203-2208: the reference leaves here, before the later checks
        e = SosProcError("The Nadal's BPDF model is not supported ==> Select another surface model")
        e.code = -6
        raise e
    if isurf == 7 and und("coef_c_maignan"):
        fail(2411, "-SURF.Maignan.C must be defined")
    if not und("tr") and p["tr"] < 0.0:
        fail(2502, "-AP.MOT must be positive or null")
    if not und("tr") and p["tr"] > 0.0:                   # (with -AP.MOT left to the Rayleigh formula the reference skips these two)
        if und("hr"):
            fail(2503, "-AP.HR (molecular scale height) must be defined")
        if p["hr"] <= 0.0:
            fail(2504, "-AP.HR must be strictly positive")
    if und("iprofil"):
        fail(2505, "-AP.AerProfile.Type must be defined")
    if int(p["iprofil"]) not in (1, 2):
        fail(2506, "-AP.AerProfile.Type must be 1 or 2")
    if int(p["iprofil"]) == 1:
        if und("ha") and p["aot_ref"] > 0.0:
            fail(2507, "-AP.AerHS.HA (aerosol scale height) must be defined")
        if p["ha"] <= 0.0 and p["aot_ref"] >= 0.0:           # as written there: an undefined HA (-999) fails even without aerosols
            fail(2508, "-AP.AerHS.HA must be strictly positive")
    if int(p["iprofil"]) == 2 and (und("zmin") or und("zmax")):
        fail(2509, "-AP.AerLayer.Zmin and -AP.AerLayer.Zmax must be defined for -AP.AerProfile.Type 2")
    if und("absprofil"):
        fail(2510, "-AP.AbsProfile.Type must be defined")
    absprofil = int(p["absprofil"])
    if not 0 <= absprofil <= 7:
        fail(2511, "-AP.AbsProfile.Type must be in 0..7")
    if absprofil == 0 and str(p["ficabsprofil"]).strip() == "NO_USER_ABS_PROFILE_FILE":
        fail(2512, "-AP.AbsProfile.UserFile must be defined for -AP.AbsProfile.Type 0")
    if int(p["iprofil"]) == 2 and absprofil != 7:
        fail(2513, "-AP.AerProfile.Type 2 requires -AP.AbsProfile.Type 7 (no gas absorption)")
    if absprofil != 7:
        if und("nustep"):
            fail(2514, "-AP.SpectralResol must be defined (1, 5 or 10 cm-1)")
        if int(p["nustep"]) not in (1, 5, 10):
            fail(25141, "-AP.SpectralResol must be 1, 5 or 10 cm-1")
        if und("imode_ckd_calcul"):
            fail(2515, "-SOS.AbsModeCKD must be defined")
    if p["igmax"] != _I and int(p["igmax"]) < 1:
        fail(2604, "-SOS.IGmax must be at least 1")
    if und("itrphi"):
        fail(2605, "-SOS.View must be defined")
    if int(p["itrphi"]) not in (1, 2):
        fail(2606, "-SOS.View must be 1 or 2")
    if int(p["itrphi"]) == 1 and und("phios"):
        fail(2607, "-SOS.View.Phi must be defined for -SOS.View 1")
    if int(p["itrphi"]) == 2:
        if und("pas_phi"):
            fail(2608, "-SOS.View.Dphi must be defined for -SOS.View 2")
        if int(p["pas_phi"]) <= 0:
            fail(2609, "-SOS.View.Dphi must be strictly positive")
    if int(p["ipolar"]) not in (0, 1):
        fail(2610, "-SOS.Ipolar must be 0 or 1")
    if (p["zout"] < 0.0 and p["zout"] != -1.0) or p["zout"] > CTE_TOA_ALT:
        fail(2611, "-SOS.OutputAlt must be -1 (standard levels) or within [0, %g] km" % CTE_TOA_ALT)


# Surface reflection matrices of the last few (surface, angles) combinations: they do not depend on the wavelength, so a
# spectrum of calls over one surface computes them once (SOS_SURFACE has the same idea with its SURFACE/ directory of files,
# SOS_SURFACE.F:260-330).  The tensors are read-only after the stream that made them has been synchronised.
_SURF_CACHE = collections.OrderedDict()
_SURF_LOCK = threading.Lock()
_SURF_CACHE_MAX = 8


def _surface_cached(key, make, device):
    import torch
    with _SURF_LOCK:
        hit = _SURF_CACHE.get(key)
        if hit is not None:
            _SURF_CACHE.move_to_end(key)
            return hit
    r = make()
    torch.cuda.current_stream(torch.device("cuda", device)).synchronize()
    with _SURF_LOCK:
        _SURF_CACHE[key] = r
        while len(_SURF_CACHE) > _SURF_CACHE_MAX:
            _SURF_CACHE.popitem(last=False)
    return r


class _Plan:
    """One wavelength between its host preparation (_prepare) and its outputs (_finish): the parameters, what the host
    derived from them, the device context and the bins of its band."""


def _trphi_azimuths(itrphi, phios, pas_phi):
    """Azimuth list (radians), table rows and PHI_FIN of the view mode (SOS_TRPHI.F:431-615)."""
    phi_fin = np.zeros(361)
    if itrphi == 1:
        phis = [math.pi + phios * math.pi / 180.0, phios * math.pi / 180.0]
        rows = [0, 1]
        phi_fin[0] = phios            # set to PHIOS+180 then overwritten by PHIOS (SOS_TRPHI.F:445,504)
    elif itrphi == 2:
        iphis = list(range(0, 361, int(pas_phi)))
        phis = [math.pi * ip / 180.0 for ip in iphis]
        rows = list(range(len(iphis)))
        phi_fin[:len(iphis)] = iphis
    else:
        raise SosProcError("-SOS.View must be 1 or 2")
    return phis, rows, phi_fin


def _zero_pages(shape):
    """A large zeroed float64 array on small pages: the fixed-size (361,81) result tables of a spectrum are mostly padding that
    is never written, and numpy asks for transparent huge pages for large blocks -- the first row written into a table would
    then clear 2 MB."""
    import mmap
    nbytes = int(np.prod(shape)) * 8
    if nbytes < (8 << 20) or not hasattr(mmap, "MADV_NOHUGEPAGE"):
        return np.zeros(shape)
    m = mmap.mmap(-1, nbytes)
    try:
        m.madvise(mmap.MADV_NOHUGEPAGE)
    except OSError:
        pass
    return np.frombuffer(m, dtype=np.float64).reshape(shape)


def _trphi_pack(n, mu, out, rows, phi_fin, block=None):
    """sosgpu_trphi's [nphi][7][W] (host array) -> PHI_FIN, THETA_FIN and the fourteen (361,81) tables.
    block: zeroed (2,7,361,81) array to fill (sos_spectrum hands out slices of one allocation per chunk of wavelengths: 3.3 MB
    blocks allocated one by one come from the heap and are cleared by hand, 0.3 ms each; one large block is fresh zero pages)."""
    theta_fin = np.zeros(81)
    theta_fin[:n] = np.degrees(np.arccos(mu))
    names = ["i", "q", "u", "sca", "ang", "rate", "lpol"]       # order of sosgpu_trphi's 7 output rows
    if block is None:
        block = np.zeros((2, 7, 361, 81))                       # the fourteen tables in one allocation
    nr = len(rows)
    if rows != list(range(nr)):
        raise ValueError("azimuth rows must be 0..nr-1")                      # (both view modes of _trphi_azimuths)
    block[0, :, :nr, :n] = out[:nr, :, n + 1:].transpose(1, 0, 2)             # up-going jj = 1..N
    block[1, :, :nr, :n] = out[:nr, :, :n][:, :, ::-1].transpose(1, 0, 2)     # down-going jj = -1..-N
    tabs = {nm: block[0, qi] for qi, nm in enumerate(names)}
    tabs_dn = {nm: block[1, qi] for qi, nm in enumerate(names)}
    return phi_fin, theta_fin, tabs, tabs_dn


def trphi_tables(ctx, rec, nf, tau, tauout, itrphi, phios, pas_phi, igli, wind, land=None):
    """SOS_TRPHI_OPTION (SOS_TRPHI.F:431-615): run the azimuth recomposition on the GPU for the azimuth list of
    the view mode and pack the fourteen (361,81) tables plus PHI_FIN(361), THETA_FIN(81)."""
    phis, rows, phi_fin = _trphi_azimuths(itrphi, phios, pas_phi)
    out = ctx.trphi(rec, nf, tau, tauout, phis, igli=igli, wind=wind, land=land).cpu().numpy()
    return _trphi_pack(ctx.n, ctx.mu, out, rows, phi_fin)


# SOS_AEROSOLS at the REFERENCE wavelength (the first of the two runs SOS_PROC makes when -SOS_Main.Wa differs from -AER.Waref,
# SOS_PROC.F:2893): it depends on the aerosol keywords only, not on the simulation wavelength, so the calls of a spectrum share
# it (the reference recomputes it -- and re-reads its MIE cache file -- in every call).  Keyed by every -AER.* keyword.
_AER_REF_CACHE = collections.OrderedDict()
_AER_REF_LOCK = threading.Lock()


def _aerosols_at_waref(p, nb_mie, os_nb, device):
    from . import aerosols as _aer
    key = tuple((k, p[k]) for _, k, _ in PARAMS if k.startswith(("imod_aer", "rn_", "in_", "igranu", "lnd_", "jd_", "imodele_", "c_wmo_",
                                                                "rh", "mode_param", "user_cv", "rtauct", "bmd_", "ficextdata",
                                                                "ficmixture", "itronc", "waref_aot", "aot_ref"))) + (
        nb_mie, os_nb, device, os.environ.get("SOS_ABS_ROOT", ""))
    with _AER_REF_LOCK:
        hit = _AER_REF_CACHE.get(key)
        if hit is not None:
            _AER_REF_CACHE.move_to_end(key)
            return dict(hit)
    out = _aer.aerosols(p, p["waref_aot"], p["aot_ref"], nb_mie, os_nb, at_waref=True, device=device)
    with _AER_REF_LOCK:
        _AER_REF_CACHE[key] = dict(out)
        while len(_AER_REF_CACHE) > 16:
            _AER_REF_CACHE.popitem(last=False)
    return out


def _validated(kw):
    """The keyword set as _prepare sees it after the parameter checks (a copy with their side effects), or None when they refuse it."""
    try:
        p = dict(kw)
        validate_parameters(p)
        return p
    except Exception:
        return None


def _gas_table_request(p):
    """The arguments of the SOS_PREPA_ABSPROFILE call _prepare will make for the validated keyword set p
    (absorption.prefetch_gas_tables), or None when the call has no gas absorption or its parameters are refused (the real pass
    reports)."""
    try:
        absprofil = int(p["absprofil"])
        if absprofil == _I or not 0 <= absprofil <= 6 or p["nustep"] == _I or p["wa_simu"] == _D:
            return None
        return (p["wa_simu"], float(p["nustep"]), p["psurf"], p["h2o"], p["o3"], p["co2"], p["ch4"], absprofil,
                str(p["ficabsprofil"]).strip())
    except Exception:
        return None


def _aerosol_call(p, aer_phase):
    """(p, nb_mie, os_nb) of the SOS_AEROSOLS run _prepare will make at the simulation wavelength for the validated keyword set p,
    or None when that step does not run or its parameters are refused (the real pass reports)."""
    try:
        if (p["aot_ref"] in (0.0, _D) or aer_phase is not None or str(p["ficuser_aer"]).strip() != "NO_USER_AEROSOLS"
                or p["waref_aot"] == _D or p["wa_simu"] in (_D, p["waref_aot"]) or p["imod_aer"] not in (0, 1, 2, 3, 4, 5)):
            return None
        nb_mie = CTE_DEFAULT_NBMU_MIE if p["nbmu_gauss_mie"] == _I else int(p["nbmu_gauss_mie"])
        os_nb = CTE_DEFAULT_OS_NB if p["nbmu_gauss_mie"] == _I else 2 * nb_mie
        return p, nb_mie, os_nb
    except Exception:
        return None


def _size_integral_requests(call, device):
    """The size integrals the aerosol run `call` (_aerosol_call) will ask for (keys of aerosols.prefetch_size_integrals)."""
    from . import aerosols as _aer
    try:
        p, nb_mie, os_nb = call
        with _aer.collect_size_integrals() as reqs:
            _aer.aerosols(p, p["wa_simu"], 0.1, nb_mie, os_nb, at_waref=False, device=device)
        return reqs
    except Exception:
        return []


# Host time of the segments of _prepare, summed over the calls (seconds; filled when SOS_PREPARE_SEGMENTS is set -- diagnostic of
# scripts/hyperspectral_bench.py): segment name -> time since the previous mark of the same call.
PREPARE_SEGMENTS = collections.OrderedDict() if os.environ.get("SOS_PREPARE_SEGMENTS") else None
_SEG_LAST = [0.0]


def _seg(name):
    if PREPARE_SEGMENTS is None:
        return True
    import time
    t = time.perf_counter()
    if name is not None:
        PREPARE_SEGMENTS[name] = PREPARE_SEGMENTS.get(name, 0.0) + t - _SEG_LAST[0]
    _SEG_LAST[0] = t
    return True


if PREPARE_SEGMENTS is not None:
    from . import solver as _solver_mod
    _solver_mod.SEG = _seg


def _prepare(kw, aer_phase=None, device=0, shard_bins=True, aer_stream=None, aer_at_wa=None):
    """Everything of one SOS_PROC call up to the CKD bin loop (SOS_PROC.F:1310-3458): parameter checks, SOS_ANGLES,
    SOS_AEROSOLS, SOS_SURFACE, SOS_PREPA_ABSPROFILE, SOS_PREPA_OS, and the profiles of every bin of the band on the device
    (SOS_ABSPROFILE + SOS_PROFILE + the rescale of SOS).  Returns a _Plan whose `ctx` the caller closes.
    shard_bins: with torch.distributed initialised, keep only this rank's slice of the band's bins (sos_proc); False: the
    whole band stays on this rank (sos_spectrum distributes wavelengths, not bins).
    aer_stream: HIP stream (torch.cuda.Stream) for the aerosol step, whose device calls are host-synchronous -- sos_spectrum keeps
    them off the streams its asynchronous work is queued on.
    aer_at_wa: the result of SOS_AEROSOLS at the simulation wavelength when sos_spectrum has formed it already (aerosols_many)."""
    _seg(None)
    missing = [k for k in SOS_PROC_KWARGS if k not in kw]
    if missing:
        raise TypeError("sos_proc() missing keyword arguments: %s" % ", ".join(missing))
    p = dict(kw)
    validate_parameters(p)
    from .solver import SosContext
    from . import surface as _surface
    from . import absorption as _abs
    import torch

    # --- parameter checks the hot path depends on (SOS_PROC.F:1310-2700 has many more; same messages' intent)
    if p["wa_simu"] == _D:
        raise SosProcError("-SOS_Main.Wa must be defined")
    if p["tetas"] == _D or not (0.0 <= p["tetas"] < 90.0):
        raise SosProcError("-ANG.Thetas must be defined in [0,90[")
    iprofil = int(p["iprofil"])
    if iprofil not in (1, 2):                                                      # SOS_PROC.F:2324-2325
        raise SosProcError("-AP.AerProfile.Type must be 1 or 2")
    if iprofil == 2 and (p["zmin"] == _D or p["zmax"] == _D):                      # :2335-2338
        raise SosProcError("-AP.AerLayer.Zmin and -AP.AerLayer.Zmax must be defined for -AP.AerProfile.Type 2")
    absprofil = int(p["absprofil"])
    if absprofil == _I or not 0 <= absprofil <= 7:
        raise SosProcError("-AP.AbsProfile.Type must be defined in 0..7")
    if p["isurf"] not in (0, 1, 2, 3, 4, 5, 6, 7):
        raise SosProcError("-SURF.Type must be in 0..7")
    if p["rho"] == _D:
        raise SosProcError("-SURF.Alb must be defined")
    if p["aot_ref"] == _D:
        raise SosProcError("-AER.AOTref must be defined")
    user_aer = str(p["ficuser_aer"]).strip()
    use_model = p["aot_ref"] != 0.0 and aer_phase is None and user_aer == "NO_USER_AEROSOLS"
    if use_model and p["imod_aer"] not in (0, 1, 2, 3, 4, 5):
        raise SosProcError("-AER.Model must be in 0..5")
    if p["hr"] == _D:
        raise SosProcError("-AP.HR must be defined")
    itrphi = p["itrphi"]
    if itrphi == 1 and p["phios"] == _D:
        raise SosProcError("-SOS.View.Phi must be defined for -SOS.View 1")
    if itrphi == 2 and p["pas_phi"] == _I:
        raise SosProcError("-SOS.View.Dphi must be defined for -SOS.View 2")
    igmax = CTE_DEFAULT_IGMAX if p["igmax"] == _I else int(p["igmax"])

    _seg('checks')
    # --- SOS_ANGLES (SOS_PROC.F:2738)
    nb_lum = CTE_DEFAULT_NBMU_LUM if p["nbmu_gauss_lum"] == _I else int(p["nbmu_gauss_lum"])
    nb_mie = CTE_DEFAULT_NBMU_MIE if p["nbmu_gauss_mie"] == _I else int(p["nbmu_gauss_mie"])
    os_nb = CTE_DEFAULT_OS_NB if p["nbmu_gauss_mie"] == _I else 2 * nb_mie          # SOS_ANGLES.F:303-312
    if p["nbmu_gauss_lum"] == _I:
        os_ns, os_nm = CTE_DEFAULT_OS_NS, CTE_DEFAULT_OS_NM
    else:
        os_ns = 2 * nb_lum
        os_nm = os_nb + os_ns                                                        # SOS_ANGLES.F:325-329
    mu, ga, n0, ind_ang = angles(nb_lum, p["tetas"], p["ficangles_user_lum"])
    n = len(mu)

    _seg('angles')
    # --- aerosols: none, a user Aerosols.txt (-AER.UserFile, SOS_PROC.F:2883-2934: SOS_AEROSOLS is not run, the file
    # is read by SOS_PREPA_OS.F:666-700), or a given expansion (stands for SOS_AEROSOLS -> Aerosols.txt)
    coef_tronca_out = None
    ta_model = None
    if use_model:
        # SOS_AEROSOLS at the reference wavelength, and again at the simulation wavelength when they differ: the optical
        # thickness scales with the extinction cross sections (SOS_PROC.F:2883-3060)
        from . import aerosols as _aer
        if p["waref_aot"] == _D:
            raise SosProcError("-AER.Waref must be defined")
        import contextlib
        try:
            with (torch.cuda.stream(aer_stream) if aer_stream is not None else contextlib.nullcontext()):
                aer_phase = _aerosols_at_waref(p, nb_mie, os_nb, device)
                ta_model = float(p["aot_ref"])
                if p["wa_simu"] != p["waref_aot"]:
                    k_ref = aer_phase["kmat1"]
                    aer_phase = (aer_at_wa if aer_at_wa is not None else
                                 _aer.aerosols(p, p["wa_simu"], 0.1, nb_mie, os_nb, at_waref=False, device=device))
                    ta_model = (aer_phase["kmat1"] / k_ref) * p["aot_ref"]
        except _aer.AerosolError as e:
            raise SosProcError(str(e), ier=-1)
        coef_tronca_out = aer_phase["coef_tronca"]
    elif p["aot_ref"] != 0.0 and aer_phase is None:
        if not os.path.exists(user_aer):
            raise SosProcError("-AER.UserFile %s not found" % user_aer)
        aer_phase = read_aerosols_file(user_aer, os_nb)
        coef_tronca_out = 0.0          # COEF_TRONCA is an output of SOS_AEROSOLS only: untouched (0) with a user file
    if p["aot_ref"] == 0.0 or aer_phase is None:
        ta = 0.0
        alpha = beta = gamma = zeta = np.zeros(os_nb + 1)
        piz, piztr, a_tronc = 0.0, 0.0, 0.0
    else:
        ta = float(p["aot_ref"]) if ta_model is None else ta_model
        alpha, beta, gamma, zeta = (np.asarray(aer_phase[k], dtype=np.float64) for k in ("alpha", "beta", "gamma", "zeta"))
        if len(beta) != os_nb + 1:
            raise SosProcError("aer_phase arrays must have OS_NB+1 = %d entries" % (os_nb + 1))
        piz, piztr, a_tronc = float(aer_phase["piz"]), float(aer_phase["piztr"]), float(aer_phase.get("a_tronc", 0.0))
        if iprofil == 1 and p["ha"] == _D:
            raise SosProcError("-AP.AerHS.HA must be defined")

    _seg('aerosols')
    # --- molecular optical thickness (SOS_PROC.F:3331-3351)
    tr = p["tr"]
    if tr == _D:
        if p["psurf"] == _D:
            raise SosProcError("-AP.MOT or -AP.Psurf must be defined")
        tr = rayleigh_optical_thickness(p["wa_simu"], p["psurf"])
    ha = p["ha"] if ta else 1.0
    zout = float(p["zout"])
    # head start: the level placement of the wavelength's no-gas profile (a ~1 ms serial chain on one wavefront) is queued now and
    # runs while the host prepares gas tables, surface and context (a refused profile is reported by make_profiles below)
    nogas = None
    if iprofil == 1:
        try:
            from .solver import nogas_profile
            nogas = nogas_profile(tr, p["hr"], ta, ha, device)
        except Exception:
            nogas = None

    # --- gas absorption: SOS_PREPA_ABSPROFILE + the weights of the CKD bins (SOS_PROC.F:3359-3416)
    use_gas = absprofil != 7
    prep = None
    if iprofil == 2 and use_gas:                                                   # SOS_PROC.F:2352
        raise SosProcError("-AP.AerProfile.Type 2 requires -AP.AbsProfile.Type 7 (no gas absorption)")
    if use_gas:
        if p["nustep"] == _I:
            raise SosProcError("-AP.SpectralResol must be defined with -AP.AbsProfile.Type != 7")
        if absprofil == 0 and str(p["ficabsprofil"]).strip() == "NO_USER_ABS_PROFILE_FILE":
            raise SosProcError("-AP.AbsProfile.UserFile must be defined with -AP.AbsProfile.Type 0")
        try:
            prep = _abs.prepa_absprofile(p["wa_simu"], float(p["nustep"]), p["psurf"], p["h2o"], p["o3"], p["co2"], p["ch4"],
                                         absprofil, str(p["ficabsprofil"]).strip())
            ik, aik, _ = _abs.bins(prep)
            xk, ro_lay = _abs.layer_tables(prep)
        except _abs.AbsorptionError as e:
            raise SosProcError(str(e), ier=-1)
    mode_ckd = int(p["imode_ckd_calcul"])
    if mode_ckd not in (1, 2):
        raise SosProcError("-SOS.AbsModeCKD must be 1 or 2")

    _seg('gas tables')
    # --- surface (SOS_PREPA_OS.F:479-497)
    isurf = int(p["isurf"])
    igli, ifresnel, imat = int(isurf == 1), int(isurf == 2), int(isurf == 1 or isurf >= 3)
    rsurf = None
    land = None
    if (isurf in (1, 2) or isurf >= 4) and p["surf_ind"] == _D:
        raise SosProcError("-SURF.Ind must be defined for sea surfaces and BPDF models")
    if isurf >= 3:
        if _D in (p["k0_roujean"], p["k1_roujean"], p["k2_roujean"]):
            raise SosProcError("-SURF.Roujean.K0/K1/K2 must be defined for -SURF.Type >= 3")
        if isurf == 6:                                    # the reference's own SOS_PROC refuses it with this message
            raise SosProcError("The Nadal's BPDF model is not supported ==> Select another surface model")
        if isurf == 7 and p["coef_c_maignan"] == _D:
            raise SosProcError("-SURF.Maignan.C must be defined for -SURF.Type 7")
        land = _surface.land_model(isurf, p["k0_roujean"], p["k1_roujean"], p["k2_roujean"], p["alpha_nadal"], p["beta_nadal"],
                                   p["coef_c_maignan"])
    lta = ta == 0.0 or piztr == 0.0                                                  # SOS.F:541-550: IBORM = 2 without aerosols
    iborm = min(2, os_nb) if lta else os_nb
    user_surf = str(p["ficsurf"]).strip()
    if imat and user_surf != "DEFAULT":
        # -SURF.File: the user's reflection matrices replace the surface computation (SOS_PROC.F:3186-3189,
        # SOS_PREPA_OS.F:605-658); the direct-beam terms of SOS_TRPHI still follow -SURF.Type
        if not os.path.exists(user_surf):
            raise SosProcError("-SURF.File %s does not exist (SOS_PREPA_OS ERROR_1020)" % user_surf, ier=-1)
        rsurf = torch.as_tensor(read_surface_file(user_surf, n, os_nb), device=torch.device("cuda", device))
    elif isurf == 1:
        if p["wind"] == _D:
            raise SosProcError("-SURF.Glitter.Wind must be defined")
        key = ("glitter", mu.tobytes(), ga.tobytes(), float(p["wind"]), float(p["surf_ind"]), os_nb, os_ns, os_nm, device)
        rsurf = _surface_cached(key, lambda: _surface.glitter_matrices(mu, ga, p["wind"], p["surf_ind"], os_nb, os_ns, os_nm,
                                                                        device=device)["rsurf"], device)
    elif land is not None:
        ind_l = p["surf_ind"] if isurf >= 4 else 1.0
        key = ("land", isurf, float(land.k0), float(land.k1), float(land.k2), float(land.alpha), float(land.beta),
               float(land.coef_c), mu.tobytes(), ga.tobytes(), float(ind_l), os_nb, os_ns, os_nm, device)
        try:
            rsurf = _surface_cached(key, lambda: _surface.land_matrices(land, mu, ga, ind_l, os_nb, os_ns, os_nm, device=device),
                                    device)
        except ValueError as e:
            raise SosProcError(str(e), ier=-1)
    if rsurf is not None and iborm < os_nb:
        rsurf = rsurf[:iborm + 1].contiguous()

    _seg('surface')
    ctx = SosContext(mu, ga, n0, alpha, beta, gamma, zeta, iborm_max=iborm, ro=p["rho"], imat_surf=imat,
                     ifresnel=ifresnel, ind_surf=p["surf_ind"] if isurf in (1, 2) or isurf >= 4 else 1.34, ron=MDF_DEFAULT,
                     ipolar=int(p["ipolar"]), igmax=igmax, rsurf=rsurf, device=device)
    _seg('context')
    pl = _Plan()
    pl.p, pl.ctx, pl.device = p, ctx, device
    pl.n, pl.mu, pl.ga, pl.n0, pl.ind_ang = n, mu, ga, n0, ind_ang
    pl.nb_lum, pl.nb_mie, pl.os_nb, pl.os_ns, pl.os_nm = nb_lum, nb_mie, os_nb, os_ns, os_nm
    pl.use_model, pl.aer_phase, pl.coef_tronca_out, pl.a_tronc = use_model, aer_phase, coef_tronca_out, a_tronc
    pl.tr, pl.ta, pl.ha, pl.use_gas, pl.prep = tr, ta, ha, use_gas, prep
    pl.itrphi, pl.igli, pl.land = itrphi, igli, land
    pl.tdifmug = None
    pl.want_trans = str(p["fictrans"]).strip() != "NO_OUTPUT"
    try:
        # --- the CKD bin loop (SOS_PROC.F:3459-3594): profiles of every bin on the device
        band_sharded = False                          # True: every rank holds a slice of the band and the partials are all-reduced
        tabs_flux = None                              # TAUABS of the band's last bin (Flux file): host array or [.][nlev] device tensor
        if not use_gas and iprofil == 1:
            # the single no-gas profile of the wavelength: SOS_PROFILE, the PROFIL-file round trip, the rescale, IBORM and the
            # output level all inside sosgpu_profile (the Python restatement profile_nogas costs 2-3 ms and stays as the checker
            # of tests/test_profile.py)
            try:
                bins = ctx.make_profiles(1, tr, p["hr"], ta, ha, None, None, a_tronc=a_tronc, piz=piz, piztr=piztr, zout=zout,
                                         absprofil=7, nogas=nogas)
            except Exception as e:
                raise SosProcError("SOS_PROFILE: %s" % e, ier=-1)
            aik = np.ones(1)
        elif not use_gas:
            h, xdel, ydel, zprof = profile_layer(tr, p["hr"], ta, float(p["zmin"]), float(p["zmax"]))
            ttot_vrai = h[-1]
            h, xdel, ydel, ib = rescale_profile(h, xdel, ydel, a_tronc, piz, piztr, os_nb)   # SOS.F:523-550
            bins = ctx.upload_bins(h[None], xdel[None], ydel[None], iborm=np.array([min(ib, iborm)], dtype=np.int32),
                                   zout=zout, zprof=zprof[None])
            if zout == -1.0:
                tauout = h[0]                                                        # SOS.F:567-568
            else:
                j = int(bins["jout"][0])
                zzv = float(bins["zz"][0])
                tauout = (1 - zzv) * h[j - 1] + zzv * h[j]                           # SOS.F:572-581
            bins["scal"] = np.array([[0.0, h[-1], ttot_vrai, tauout]])
            aik = np.ones(1)
        else:
            # Several GPUs (torch.distributed initialised, one process per GPU): the bins of the band are dealt to the ranks
            # by cost (dist.balanced_shards); solve_band's one all-reduce joins them and every rank returns the same 23
            # outputs.  A rank may hold no bin at all.  (-SOS.AbsModeCKD 2 has a single bin: not sharded.)
            rank, world = _dist_rank_world() if shard_bins else (0, 1)
            band_sharded = world > 1 and mode_ckd != 2
            tabs_last = None
            if band_sharded:
                from . import dist as _dist
                tabs_last = ctx.absorption_profiles(ik[-1:], xk, ro_lay)             # TAUABS of the band's last bin (Flux file)
                mine = _dist.balanced_shards(_abs.bin_costs(ik, xk, ro_lay, tr + ta), world)[rank]
                ik, aik = ik[mine], aik[mine]
            tabs = ctx.absorption_profiles(ik, xk, ro_lay) if len(aik) else None     # SOS_ABSPROFILE of every bin of this rank
            if mode_ckd == 2:
                # one profile from the band-mean transmission of every level (SOS_PROC.F:3609-3676)
                tb = tabs.cpu().numpy()
                trs = np.zeros(tb.shape[1])
                for b in range(tb.shape[0]):
                    trs = trs + aik[b] * np.exp(-tb[b])
                tabs = np.maximum(-np.log(trs), 0.0)[None]
                aik = np.ones(1)
            try:
                if len(aik):
                    bins = ctx.make_profiles(len(aik), tr, p["hr"], ta, ha, prep["altabs"], tabs, a_tronc=a_tronc, piz=piz,
                                             piztr=piztr, zout=zout, absprofil=absprofil, nogas=nogas)
                else:
                    bins = dict(nb=0, scal=None)
            except Exception as e:
                raise SosProcError("SOS_PROFILE: %s" % e, ier=-1)
            tabs_flux = tabs_last if band_sharded else tabs
        if pl.want_trans and bins["nb"]:                                             # SOS.F:600-635
            tdifmus_b, pl.tdifmug = ctx.diffuse_transmissions(bins)
            sc = bins["scal"] if hasattr(bins["scal"], "clone") else torch.from_numpy(np.asarray(bins["scal"])).to(ctx.device)
            sc = sc.clone()
            sc[:, 0] = tdifmus_b
            bins["scal"] = sc
    except BaseException:
        ctx.close()
        raise
    _seg('profiles')
    pl.bins, pl.aik, pl.band_sharded, pl.tabs_flux = bins, aik, band_sharded, tabs_flux
    return pl


def _trphi_launch(pl, rec0, nf, tau, tauout):
    """Queue SOS_TRPHI for the azimuths of the view mode; returns the device tensor [nphi][7][W] (no synchronisation)."""
    phis, pl.rows, pl.phi_fin = _trphi_azimuths(pl.itrphi, pl.p["phios"], pl.p["pas_phi"])
    return pl.ctx.trphi(rec0, nf, tau, tauout, phis, igli=pl.igli, wind=pl.p["wind"] if pl.igli else 0.0, land=pl.land)


def _finish(pl, out, rec0, fin, g=0, block=None):
    """The tail of SOS_PROC (SOS_PROC.F:3755-3874) on the host: the (361,81) tables from sosgpu_trphi's output `out` (host
    array), fluxes, result files (rank 0 only), the 23-tuple.  rec0: aggregated records [S][3][W] of this wavelength
    (device); fin: dist.finish_scalars dictionary, g the wavelength's segment in it."""
    p, n, mu = pl.p, pl.n, pl.mu
    nf = int(fin["n_orders"][g])
    tau_agg, ttot_vrai_agg = float(fin["ttot_tronc"][g]), float(fin["ttot_vrai"][g])
    phi_fin, theta_fin, up, dn = _trphi_pack(n, mu, out, pl.rows, pl.phi_fin, block)
    emoins, eplus = float(fin["emoins"][g]), float(fin["eplus"][g])
    from . import absorption as _abs
    resbin = str(p["ficsos_res_bin"]).strip()
    resroot = str(p["resroot"]).strip() if getattr(pl, "writes_files", _dist_rank_world()[0] == 0) else ""
    if resroot:                                   # SOS_PROC.F:1342-1500: results under RESROOT/SOS
        os.makedirs(os.path.join(resroot, "SOS"), exist_ok=True)
        # the two angle files SOS_ANGLES always writes (SOS_ANGLES.F:367-376, 494-506)
        write_mie_angles(os.path.join(resroot, "SOS", str(p["ficangles_res_mie"]).strip()), pl.nb_mie, pl.os_nb,
                         str(p["ficangles_user_mie"]).strip())
        write_used_angles(os.path.join(resroot, "SOS", str(p["ficangles_res_lum"]).strip()), mu, pl.ga, pl.n0, pl.ind_ang,
                          pl.nb_lum, p["tetas"], pl.os_nb, pl.os_ns, pl.os_nm, str(p["ficangles_user_lum"]).strip())
        if pl.use_model:
            write_aerosols_file(os.path.join(resroot, "SOS", str(p["ficgranu"]).strip()), pl.aer_phase, pl.aer_phase["kmat1"],
                                pl.aer_phase["kmat2"])
        elif p["aot_ref"] == 0.0:                 # SOS_AEROSOLS writes an all-zero file for an aerosol-free run
            z = np.zeros(pl.os_nb + 1)
            write_aerosols_file(os.path.join(resroot, "SOS", str(p["ficgranu"]).strip()),
                                dict(alpha=z, beta=z, gamma=z, zeta=z, a_tronc=0.0, piztr=0.0, piz=0.0))
        write_result_bin(os.path.join(resroot, "SOS", resbin), rec0[:nf].cpu().numpy())
    cs = math.cos(math.pi * p["tetas"] / 180.0)
    tdir_tronc = math.exp(-tau_agg / cs)                                          # SOS_PROC.F:3831-3837
    tdir_vrai = math.exp(-ttot_vrai_agg / cs)
    flux_diff_down = emoins + tdir_tronc - tdir_vrai
    flux_down = emoins + tdir_tronc
    if resroot and pl.want_trans:
        write_trans_file(os.path.join(resroot, "SOS", str(p["fictrans"]).strip()), p["tetas"], mu, tau_agg, ttot_vrai_agg,
                         float(fin["tdifmus"][g]), fin["tdifmug"][g])
    if resroot and str(p["ficflux"]).strip() != "NO_OUTPUT":
        tl = pl.tabs_flux
        tabs_flux = np.zeros(_abs.NLEVEL) if tl is None else (tl[-1].cpu().numpy() if hasattr(tl, "cpu") else np.asarray(tl)[-1])
        zal = pl.prep["userprofil"][:, 0] if pl.use_gas else np.linspace(0., 0., _abs.NLEVEL)
        write_flux_file(os.path.join(resroot, "SOS", str(p["ficflux"]).strip()), p["tetas"], tdir_vrai, flux_diff_down, flux_down,
                        eplus, zal, pl.tr, p["hr"], pl.ta, pl.ha, tabs_flux)
    ind_angout = np.zeros(81, dtype=np.int32)
    ind_angout[:n] = pl.ind_ang
    return (n, ind_angout, phi_fin, theta_fin,
            up["sca"], up["i"], up["q"], up["u"], up["ang"], up["rate"], up["lpol"],
            dn["sca"], dn["i"], dn["q"], dn["u"], dn["ang"], dn["rate"], dn["lpol"],
            tdir_vrai, flux_diff_down, flux_down, eplus, pl.a_tronc if pl.coef_tronca_out is None else pl.coef_tronca_out)


def sos_proc(aer_phase=None, device=0, **kw):
    """Drop-in for `sos.sos_proc(**kwargs)` (f2py of SOS_PROC, SOS_PROC.F:415) on the MI355X hot path.
    Returns the reference's 23-tuple (names in OUTPUT_NAMES).

    aer_phase (extension, not a reference keyword): dict(alpha, beta, gamma, zeta [OS_NB+1 each], piz, piztr,
    a_tronc) -- the content of the reference's Aerosols.txt when `-AER.AOTref` > 0 (bypasses the Mie / size-distribution
    step).  With torch.distributed initialised the call is a collective: the band's CKD bins are sharded over the ranks
    (one all-reduce) and every rank returns the same outputs."""
    from .solver import SosBinError
    rank, world = _dist_rank_world()
    pl, err = None, None
    try:
        pl = _prepare(kw, aer_phase, device, shard_bins=True)
    except Exception as e:                         # noqa: BLE001 -- re-raised below, after the ranks have agreed
        err = e
    if world > 1:
        # A failure that strikes one rank only (device memory, a HIP error) must not leave the others waiting in the band's
        # all-reduce: the ranks agree on an error flag first (one integer, MAX).  Parameter errors are the same on every rank.
        bad = _any_rank_failed(err is not None, device)
        if bad and err is None:
            pl.ctx.close()
            raise SosProcError("sos_proc: another rank failed while preparing this call", ier=-1)
    if err is not None:
        raise err
    try:
        # one fused solve of the band's bins, one aggregate (+ the all-reduce of a sharded band)
        try:
            rec, fin = pl.ctx.solve_band(pl.bins, pl.aik, tdifmug=pl.tdifmug, reduce=pl.band_sharded)
        except SosBinError as e:
            raise SosProcError(str(e), ier=-1)
        out = _trphi_launch(pl, rec[0], int(fin["n_orders"][0]), float(fin["ttot_tronc"][0]), float(fin["tauout"][0]))
        return _finish(pl, out.cpu().numpy(), rec[0], fin, 0)
    finally:
        pl.ctx.close()


def _any_rank_failed(failed, device):
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=torch.device("cuda", device) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def sos_proc_many(kwargs_list, n_workers=8, device=0):
    """A spectrum of independent sos_proc calls (one per wavelength: the reference runs them one after the other,
    binding/run_sos.py:640): the list interface of sos_spectrum -- identical outputs to the sequential loop, bit for bit.
    Round 2 issued the calls from `n_workers` host threads with a HIP stream each; the Python host work under the interpreter
    lock bounded that form at 1.3 x the plain loop (and below it once the calls became cheaper).  Now: one host thread, the
    preparation kernels of the wavelengths spread over `n_workers` streams, all bins of all wavelengths in one launch per
    kernel variant (sos_spectrum).  Give each call its own `-SOS_Main.ResRoot` when result files are wanted (the file names
    inside are fixed, as in the reference).  Export GPU_MAX_HW_QUEUES=16 before the first GPU call.  Returns the list of
    23-tuples in order; a failing call raises its exception."""
    return sos_spectrum(kwargs_list, device=device, prep_streams=max(1, int(n_workers)))


# ---------------------------------------------------------------------------------------------------------
# A whole spectrum through the drop-in (BASELINE config 5: hyperspectral runs)
# ---------------------------------------------------------------------------------------------------------
_TABLE_NAMES = (4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17)     # the fourteen (361,81) tables of the 23-tuple


def _compact_outputs(t, nrow):
    """A 23-tuple without the zero padding of its fixed-size tables (what travels between ranks): the (361,81) tables cut to
    the `nrow` azimuth rows and N columns in use."""
    n = int(t[0])
    return tuple(np.ascontiguousarray(x[:nrow, :n]) if i in _TABLE_NAMES else x for i, x in enumerate(t)) + (nrow,)


def _expand_outputs(c, block=None):
    """block: zeroed (14, 361, 81) array for the tables (a slice of one allocation per gather, _zero_pages)."""
    nrow = c[-1]
    out = []
    t = 0
    for i, x in enumerate(c[:-1]):
        if i in _TABLE_NAMES:
            full = np.zeros((361, 81)) if block is None else block[t]
            t += 1
            full[:nrow, :x.shape[1]] = x
            out.append(full)
        else:
            out.append(x)
    return tuple(out)


def _gather_results(results, mine, nrows, world):
    """Every rank receives the 23-tuples of the wavelengths the other ranks computed (all_gather_object of the compacted
    tuples; the only exchange of a wavelength-partitioned spectrum)."""
    import torch.distributed as dist
    part = [(i, _compact_outputs(results[i], nrows[i])) for i in mine]
    parts = [None] * world
    dist.all_gather_object(parts, part)
    todo = [(i, c) for pr in parts for i, c in pr if results[i] is None]
    blocks = _zero_pages((len(todo), len(_TABLE_NAMES), 361, 81)) if todo else None
    for k, (i, c) in enumerate(todo):
        if results[i] is None:
            results[i] = _expand_outputs(c, blocks[k])


def spectrum_costs(kwargs_list):
    """Relative cost of every call of a spectrum, from its keywords alone (no aerosol, surface or profile work): number of CKD
    bins of the spectral interval (product of the gases' exponential-term counts, read from the CKD table headers -- 1 without
    gas absorption or in -SOS.AbsModeCKD 2) x dist.bin_cost of the wavelength's scattering optical depth (Rayleigh formula +
    -AER.AOTref) and a moderate gas column.  Used to deal the wavelengths to the ranks."""
    from . import absorption as _abs
    from .dist import bin_cost
    costs = np.zeros(len(kwargs_list))
    for i, kw in enumerate(kwargs_list):
        nb = 1
        try:
            if int(kw["absprofil"]) != 7 and int(kw["imode_ckd_calcul"]) == 1:
                nb = _abs.band_bin_count(kw["wa_simu"], float(kw["nustep"]))
        except Exception:               # a call that will fail in validation costs nothing; the error is raised by its owner
            nb = 1
        tr = kw.get("tr", _D)
        if tr == _D:
            tr = rayleigh_optical_thickness(kw["wa_simu"], kw["psurf"]) if kw.get("psurf", _D) != _D and kw.get("wa_simu", _D) != _D else 0.1
        ta = kw.get("aot_ref", 0.0)
        ta = 0.0 if ta == _D else ta
        norders = 3 if ta == 0.0 else 40          # Fourier orders: 3 for a molecular atmosphere (IBORM = 2), tens with aerosols
        costs[i] = nb * float(bin_cost(tr + ta, 0.5 if nb > 1 else 0.0)) * norders
    return costs


def sos_spectrum(kwargs_list, aer_phases=None, device=0, gather=True, chunk=256, timings=None, prep_streams=16, parts=4):
    """A spectrum of sos_proc calls -- one per wavelength, as the reference issues them one after the other
    (binding/run_sos.py:640-695; the bin loop of each is SOS_PROC.F:3459-3594) -- as ONE pass over the GPU:

      1. the host preparation of every wavelength (parameter checks, angles, aerosol model, gas tables, surface) and the
         profiles of all its CKD bins on the device (sosgpu_absprofile + sosgpu_profile), nothing waited for;
      2. ALL bins of ALL wavelengths in one launch of the fused solver per group of wavelengths sharing a kernel variant
         (direction count, highest Fourier order, surface-matrix flag, level capacity, output level): a device table of the
         wavelength contexts, every bin carrying the index of its own (sosgpu_ctx_table + sosgpu_os_solve_multi), and one
         segmented SOS_AGGREGATE;
      3. one device-to-host copy of the band scalars, the azimuth recompositions (sosgpu_trphi) of every wavelength queued
         back to back, one copy of their results, and the 23-tuples on the host.

    Outputs are those of `[sos_proc(**kw) for kw in kwargs_list]`, bit for bit (the table form of the kernels computes the same
    instruction sequence per bin; bands up to 128 bins are aggregated in the reference's serial bin order in both).
    Wavelengths are processed `chunk` at a time (their source operators are resident together: 32 MB each at N = 41).
    Calls that need the diffuse transmissions of -SOS.Trans run through the per-wavelength path.

    With torch.distributed initialised (one process per GPU) the WAVELENGTHS are dealt to the ranks by cost
    (spectrum_costs, dist.balanced_shards) -- no bin of a band leaves its rank, so there is no all-reduce, only the gather of
    the results ("partition by wavelength first"): gather=True returns the full list on every rank
    (all_gather_object of the compacted tuples), gather=False returns None in the slots of other ranks.  Result files of a
    call (-SOS_Main.ResRoot) are written by the rank that owns it.
    aer_phases: optional list parallel to kwargs_list of `aer_phase` dictionaries (see sos_proc) or None.
    timings: optional dict, filled with host-side phase times in seconds (prepare, solve_launch, wait, trphi, finish).
    parts: a chunk is handed to the solver in this many parts (the solves of one part overlap the host preparation of the next).
    prep_streams: HIP streams the per-wavelength preparation kernels are spread over (export GPU_MAX_HW_QUEUES=16 to give them
    hardware queues of their own, solver.solve_many).  The preparation kernels of one wavelength are a serial chain of about
    3 ms on a few wavefronts (level placement of the no-gas profile and of every bin: bisections), so the device side alone
    sustains prep_streams / 3 ms wavelengths per second: 16 streams keep it ahead of a host that spends 0.6 ms per wavelength."""
    import time
    import torch
    from . import capi
    from .solver import ContextTable, SosBinError, _upload, concat_bins, solve_spectrum
    from . import dist as _dist
    from . import aerosols as _aer
    from . import absorption as _abs
    capi.lib()
    nwl = len(kwargs_list)
    if aer_phases is None:
        aer_phases = [None] * nwl
    if len(aer_phases) != nwl:
        raise ValueError("aer_phases must be parallel to kwargs_list")
    rank, world = _dist_rank_world()
    if world > 1:
        mine = [int(i) for i in _dist.balanced_shards(spectrum_costs(kwargs_list), world)[rank]]
    else:
        mine = list(range(nwl))
    results = [None] * nwl
    debug = bool(os.environ.get("SOS_SPECTRUM_DEBUG"))
    nrows = {}
    tm = dict(prepare=0.0, solve_launch=0.0, wait=0.0, trphi=0.0, finish=0.0)
    dev = torch.device("cuda", device)
    main_st = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(max(1, int(prep_streams)))]
    aer_st = torch.cuda.Stream(device=dev)
    if len(mine) >= 64 and "GPU_MAX_HW_QUEUES" not in os.environ and not getattr(sos_spectrum, "_warned_queues", False):
        import warnings
        sos_spectrum._warned_queues = True
        warnings.warn("GPU_MAX_HW_QUEUES is not set: the HIP runtime maps the %d preparation streams of sos_spectrum onto 4 hardware "
                      "queues and the device side of the preparation becomes the limit (about 2/3 of the throughput); export "
                      "GPU_MAX_HW_QUEUES=16 before the first GPU call of the process (16 // processes when several processes share "
                      "the GPU)" % len(side), RuntimeWarning, stacklevel=2)
    # The cyclic garbage collector is paused for the pass: a spectrum allocates tens of container objects per wavelength next to
    # a growing list of result tuples, and the collections this triggers re-walk the results again and again (SOS_SPECTRUM_KEEP_GC=1
    # leaves the collector alone).  Nothing here relies on it: contexts are closed explicitly, tensors are freed by reference count.
    import gc
    pause_gc = gc.isenabled() and not os.environ.get("SOS_SPECTRUM_KEEP_GC")
    if pause_gc:
        gc.disable()
    try:
        for c0 in range(0, len(mine), max(1, int(chunk))):
            idx = mine[c0:c0 + max(1, int(chunk))]
            plans = []
            try:
                t0 = time.perf_counter()
                # the preparation of wavelength k queues its device work (source operators, absorption and level profiles of its
                # bins: latency-bound kernels of 0.1-2 ms on a few wavefronts) on side stream k mod n: the wavelengths overlap on the
                # device, and the launches below wait for all of them
                for st in side + [aer_st]:
                    st.wait_stream(main_st)
                # the size-distribution integrals of the chunk's wavelengths, queued ahead (aerosols.prefetch_size_integrals)
                valid = {i: v for i, v in ((i, _validated(kwargs_list[i])) for i in idx) if v is not None}
                acalls = {i: c for i, c in ((i, _aerosol_call(v, aer_phases[i])) for i, v in valid.items()) if c is not None}
                reqs = []
                for c in acalls.values():
                    reqs += _size_integral_requests(c, device)
                if reqs:
                    with torch.cuda.stream(aer_st):
                        _aer.prefetch_size_integrals(reqs)
                # ... and the gas tables of the chunk's wavelengths, interpolated to the layers in one pass
                greqs = [r for r in (_gas_table_request(v) for v in valid.values()) if r is not None]
                if greqs:
                    _abs.prefetch_gas_tables(greqs)
                # ... and SOS_AEROSOLS at the simulation wavelength of each, the Legendre expansions formed together (the gas tables
                # above were made while the size integrals ran)
                aer_wa = {}
                by_angles = collections.OrderedDict()
                for i, (pp, nbm, nbo) in acalls.items():
                    by_angles.setdefault((nbm, nbo), []).append((i, pp))
                for (nbm, nbo), members in by_angles.items():
                    with torch.cuda.stream(aer_st):
                        res = _aer.aerosols_many([(pp, pp["wa_simu"], 0.1) for _, pp in members], nbm, nbo, device=device)
                    for (i, _), r in zip(members, res):
                        if r is not None:
                            aer_wa[i] = r
                # The chunk goes to the solver in a few parts: the solves of a part run on the device while the host prepares the
                # next one, so that only the last part's solve is waited for below.
                solved = []                       # (plans, rec [nw][S][3][W], scal [nw][10+N]) device tensors
                # (at least 32 wavelengths per part -- a launch per kernel variant each; SOS_SPECTRUM_MIN_PART: the tests' override)
                nsub = max(1, min(int(parts), len(idx) // max(1, int(os.environ.get("SOS_SPECTRUM_MIN_PART", "32")))))
                step = max(1, -(-len(idx) // nsub))
                for s0 in range(0, len(idx), step):
                    part = []
                    for k, i in enumerate(idx[s0:s0 + step], s0):
                        with torch.cuda.stream(side[k % len(side)]):
                            pl = _prepare(kwargs_list[i], aer_phases[i], device, shard_bins=False, aer_stream=aer_st,
                                          aer_at_wa=aer_wa.get(i))
                        if debug:
                            torch.cuda.synchronize(dev)
                            print("[sos_spectrum] prepared", i, flush=True)
                        pl.writes_files = True
                        pl.index = i
                        plans.append(pl)
                        part.append(pl)
                    for st in side:
                        main_st.wait_stream(st)
                    t1 = time.perf_counter()
                    tm["prepare"] += t1 - t0
                    # --- groups of wavelengths one launch can cover
                    groups = collections.OrderedDict()
                    single = []
                    for pl in part:
                        b = pl.bins
                        if pl.tdifmug is not None or b["nb"] == 0 or not isinstance(b.get("scal"), torch.Tensor):
                            single.append(pl)
                            continue
                        key = (pl.n, pl.ctx.smax, pl.ctx.os_nb, bool(pl.ctx._rsurf is not None), b["lp"], b["jout"] is not None)
                        groups.setdefault(key, []).append(pl)
                    for key, gp in groups.items():
                        if len(gp) == 1:
                            single.append(gp[0])
                            continue
                        table = ContextTable([pl.ctx for pl in gp])
                        bins, cob, seg = concat_bins([pl.bins for pl in gp])
                        aik = _upload(torch.from_numpy(np.concatenate([np.asarray(pl.aik, dtype=np.float64) for pl in gp])), dev)
                        if debug:
                            print("[sos_spectrum] group", key, "wavelengths", [pl.index for pl in gp], "bins", bins["nb"], flush=True)
                        rec, scal = solve_spectrum(table, bins, cob, seg, aik, order=None)
                        if debug:
                            torch.cuda.synchronize(dev)
                            print("[sos_spectrum]   done", flush=True)
                        solved.append((gp, rec, scal, table))
                    for pl in single:
                        out = pl.ctx.solve(pl.bins, pl.ctx.alloc_outputs(pl.bins["nb"], zero=False))
                        rec, scal = pl.ctx.aggregate(out, pl.aik, scal=pl.bins.get("scal"), tdifmug=pl.tdifmug)
                        solved.append(([pl], rec, scal, None))
                    t0 = time.perf_counter()
                    tm["solve_launch"] += t0 - t1
                t2 = time.perf_counter()
                # --- one copy of all band scalars (waits for the solves), then every azimuth recomposition back to back
                scal_all = torch.cat([s.reshape(-1) for _, _, s, _ in solved]).cpu().numpy()
                t3 = time.perf_counter()
                tm["wait"] += t3 - t2
                outs, pos = [], 0
                for gp, rec, scal, _ in solved:
                    sw = scal.shape[1]
                    fin = _dist.finish_scalars(scal_all[pos:pos + len(gp) * sw].reshape(len(gp), sw))
                    pos += len(gp) * sw
                    for g, pl in enumerate(gp):
                        if fin["min_orders"][g] < 0:
                            raise SosProcError("SOS_OS: wavelength %d (%r microns) holds a malformed bin (NT outside 1..CTE_OS_NT or "
                                               "IBORM out of range)" % (pl.index, pl.p["wa_simu"]), ier=-1)
                        pl.fin, pl.g, pl.rec0 = fin, g, rec[g]
                        outs.append(_trphi_launch(pl, rec[g], int(fin["n_orders"][g]), float(fin["ttot_tronc"][g]),
                                                  float(fin["tauout"][g])))
                flat = torch.cat([o.reshape(-1) for o in outs]).cpu().numpy()
                t4 = time.perf_counter()
                tm["trphi"] += t4 - t3
                pos, k = 0, 0
                blocks = _zero_pages((len(outs), 2, 7, 361, 81))  # the result tables of the chunk's wavelengths (views of it)
                for gp, _, _, _ in solved:
                    for pl in gp:
                        cnt = outs[k].numel()
                        results[pl.index] = _finish(pl, flat[pos:pos + cnt].reshape(outs[k].shape), pl.rec0, pl.fin, pl.g, blocks[k])
                        nrows[pl.index] = len(pl.rows)
                        pos += cnt
                        k += 1
                tm["finish"] += time.perf_counter() - t4
            finally:
                _aer.drop_prefetched_size_integrals()
                _abs.drop_prefetched_gas_tables()
                for st in side + [aer_st]:
                    st.synchronize()
                main_st.synchronize()                             # the table launches read every context's operators
                for pl in plans:
                    pl.ctx.close()
    finally:
        if pause_gc:
            gc.enable()
    if timings is not None:
        timings.update(tm)
    if world > 1 and gather:
        _gather_results(results, mine, nrows, world)
    return results


def write_trans_file(path, tetas, mu, ttot_tronc, ttot_vrai, tdifmus, tdifmug):
    """-SOS.Trans file (SOS_PROC.F:3785-3822, formats 1005, 1006, 1010, 2010): direct transmission for the true optical
    depth, diffuse transmissions brought back to the true atmosphere by + exp(-tau_tr/mu) - exp(-tau/mu)."""
    cs = math.cos(math.pi * tetas / 180.0)
    with open(path, "w") as f:
        f.write("Solar Zenith Angle : %7.3f\n" % tetas)
        f.write("Direct transmission TOA -> surface : %8.4f\n" % math.exp(-ttot_vrai / cs))
        f.write("  \n")
        f.write(" Diffuse transmittance : TOA -> surface\n")
        f.write("    thetas = %6.3f   td(thetas) = %7.4f\n" % (tetas, tdifmus + math.exp(-ttot_tronc / cs) - math.exp(-ttot_vrai / cs)))
        f.write("  \n")
        f.write(" Diffuse transmittance : surface -> TOA\n")
        for j in range(len(mu)):
            td = tdifmug[j] + math.exp(-ttot_tronc / mu[j]) - math.exp(-ttot_vrai / mu[j])
            f.write("    thetav = %6.3f   td(thetav) = %7.4f\n" % (math.degrees(math.acos(mu[j])), td))


def write_flux_file(path, tetas, tdir_vrai, flux_diff_down, flux_down, eplus, zalt, tr, hr, ta, ha, tauabs):
    """-SOS.Flux file (SOS_PROC.F:3840-3874, formats 1005, 2016-2020)."""
    with open(path, "w") as f:
        f.write("Solar Zenith Angle : %7.3f\n" % tetas)
        f.write("  \n")
        f.write(" Downward fluxes at BOA (normalized by TOA solar flux)\n")
        f.write("   - Downward direct flux at BOA : %9.5f\n" % tdir_vrai)
        f.write("   - Downward diffuse flux at BOA: %9.5f\n" % flux_diff_down)
        f.write("   ==> Downward total flux at BOA: %9.5f\n" % flux_down)
        f.write("  \n")
        f.write(" Upward diffuse flux at TOA (normalized by TOA solar flux): %s\n" % repr(eplus))
        f.write("\n\n")
        f.write(" According to the following profile\n")
        f.write(" Z(km)    MOT     AOT     GOT     TOTAL\n")
        nl = len(zalt)
        for i in range(nl - 1, -1, -1):                       # I = CTE_ABS_NBLEV .. 1; TAUABS(CTE_ABS_NBLEV+1-I)
            z = zalt[i]
            trz, taz, tgz = tr * math.exp(-z / hr), ta * math.exp(-z / ha), tauabs[nl - 1 - i]
            f.write("%7.2f  %7.4f %7.4f %7.4f %7.4f\n" % (z, trz, taz, tgz, trz + taz + tgz))


def write_result_bin(path, rec):
    """Write aggregated Fourier records [F][3][W] (I,Q,U) as the reference's SOS_Result.bin: Fortran sequential
    unformatted, one record per order holding Q(-N:N), U(-N:N), I(-N:N) (SOS_OS.F:1572-1574)."""
    rec = np.asarray(rec, dtype="<f8")
    with open(path, "wb") as f:
        for s in range(rec.shape[0]):
            payload = np.concatenate([rec[s, 1], rec[s, 2], rec[s, 0]]).tobytes()
            m = np.array([len(payload)], "<i4").tobytes()
            f.write(m + payload + m)


# ---------------------------------------------------------------------------------------------------------
# SOS_Up.txt / SOS_Down.txt as the reference's command-line program writes them (SOS_ABS_MAIN.F:2250-2444)
# ---------------------------------------------------------------------------------------------------------
_SEP102 = "#" + "-" * 101
# (text, width of the `(Aw)` edit descriptor it is written with): the Fortran source is UTF-8, so the degree sign counts
# two bytes and the two "180 deg / 0 deg" lines lose their last character -- part of the file format as shipped
_HDR_COLS = [("#   VZA     :  Viewing Zenith Angle (in degrees)", 48), ("#   SCA_ANG :  Scattering angle (in degrees)", 44),
             ("#   I       :  Stokes parameter I at output altitude z (in sr-1)", 64),
             ("#              normalised to the extraterrestrial solar irradiance (PI * L(z) / Esun)", 85),
             ("#   Q       :  Stokes parameter Q at output altitude z (in sr-1)", 64),
             ("#              normalised to the extraterrestrial solar irradiance", 66),
             ("#   U       :  Stokes parameter U at output altitude z (in sr-1)", 64),
             ("#              normalised to the extraterrestrial solar irradiance ", 66),
             ("#   POL_ANG :  Polarization angle (in degrees).  Note: if undefined the value is -999.00", 88),
             ("#   POL_RATE:  Degree of polarization (in %)", 44),
             ("#   IPOL    :  Polarized intensity at level z (in sr-1)", 55),
             ("#              normalised to the extraterrestrial solar irradiance (PI * Lpol(z) / Esun)", 88)]
_SAT_180 = ("#        180° <-> Satellite and Sun in the same half-plane", 58)
_SAT_0 = ("#          0° <-> Satellite and Sun in opposite half-planes with respect to the zenith direction", 96)


def _fmt_a(text, width):
    """Fortran `(Aw)` of a character constant: the first w BYTES of the UTF-8 source text when it is longer, right-justified in w
    columns when shorter; list-directed trailing blanks are kept by the runtime, not added here (flang trims none)."""
    b = text.encode("utf-8")
    return (b" " * (width - len(b)) + b) if len(b) < width else b[:width]


def main_output_header(itrphi, updown, zalt, phios=0.0):
    """SOS_OUTPUT_HEADER (fixed azimuth, SOS_TRPHI.F:1570-1680) / SOS_OUTPUT_HEADER_POLAR_DIAG (:1705-1796) as bytes."""
    up = updown == 1
    lines = []
    if itrphi == 1:
        lines.append(_fmt_a("# UPWARD RADIANCE FIELD VERSUS THE VIEWING ZENITH ANGLE", 55) if up else
                     _fmt_a("# DOWNWARD RADIANCE FIELD VERSUS THE VIEWING ZENITH ANGLE", 57))
        lines += [_fmt_a("# (RELATIVE AZIMUTH AND ALTITUDE ARE FIXED)", 43), _fmt_a(_SEP102, 102),
                  _fmt_a("# Relative azimuth (degrees) :", 30), b"#", _fmt_a("#      Relative azimuth convention :", 36)]
        if up:
            lines += [_fmt_a(*_SAT_180), _fmt_a(*_SAT_0)]
        else:
            lines += [_fmt_a("#        180° <-> Viewing direction and Sun in the same half-plane", 66),
                      _fmt_a("#          0° <-> Viewing direction and Sun in opposite half-planes with respect to the zenith "
                             "direction", 104)]
        lines += [b"#", _fmt_a("#      Simulated relative azimuth (degrees) :", 45),
                  _fmt_a("#          for VZA < 0 (sign convention):", 41) + b" " + ("%7.3f" % (phios + 180.0)).encode(),
                  _fmt_a("#          for VZA > 0 (sign convention):", 41) + b" " + ("%7.3f" % phios).encode(), b"#"]
    else:
        lines.append(_fmt_a("#UPWARD RADIANCE FIELD VERSUS THE AZIMUTH ANGLE AND VIEWING ZENITH ANGLE", 72) if up else
                     _fmt_a("#DOWNWARD RADIANCE FIELD VERSUS THE AZIMUTH ANGLE AND VIEWING ZENITH ANGLE", 74))
        lines += [_fmt_a("#(ALTITUDE FIXED)", 17), _fmt_a(_SEP102, 102), _fmt_a("# Relative azimuth convention :", 31),
                  _fmt_a(*_SAT_180), _fmt_a(*_SAT_0), b"#"]
    lines += [_fmt_a("# Value of the selected altitude for the output (km) :", 54) + b" " + ("%7.3f" % zalt).encode(), b"#",
              _fmt_a("# Columns parameters :", 22)]
    if itrphi != 1:
        lines.append(_fmt_a("#   PHI     :  Relative azimuth Angle (in degrees)", 50))
    lines += [_fmt_a(t, w) for t, w in _HDR_COLS]
    lines.append(_fmt_a(_SEP102, 102))
    if itrphi == 1:
        lines += [_fmt_a("#   VZA     SCA_ANG       I              Q              U         POL_ANG  POL_RATE    IPOL", 91),
                  _fmt_a("#(degrees) (degrees)    (sr-1)         (sr-1)         (sr-1)      (degrees)  (%)      (sr-1)", 92)]
    else:
        lines += [_fmt_a("#   PHI      VZA     SCA_ANG        I              Q              U       POL_ANG  POL_RATE    IPOL", 99),
                  _fmt_a("#(degrees) (degrees) (degrees)    (sr-1)         (sr-1)         (sr-1)    (degrees)  (%)      (sr-1)", 100)]
    return b"".join(ln + b"\n" for ln in lines)


def _f7_2(x):
    s = "%7.2f" % x
    return s if len(s) == 7 else "*" * 7                       # Fw.d overflow


def write_main_output(path_up, path_down, outputs, itrphi, phios, pas_phi, zout, user_up=None, user_down=None):
    """The two ASCII result files of the reference's command-line program, byte for byte: headers of SOS_OUTPUT_HEADER*,
    then format 55 `2(2X,F7.2),2X,3(E13.6,2X),2(F7.2,2X),E13.6` (fixed azimuth: the phi + 180 half plane with negated zenith
    angles in reversed order, then the phi half plane, SOS_ABS_MAIN.F:2304-2405) or format 56
    `3(2X,F7.2),1X,3(E13.6,2X),2(1X,F7.2),E13.6` (polar diagram, :2450-2496).  outputs: the 23-tuple of sos_proc.
    user_up / user_down: optional paths of the -SOS.ResFileUp.UserAng / -SOS.ResFileDown.UserAng files (user angles only,
    IND_ANGOUT = 1).  One quirk of the reference is kept: in the polar diagram the SCA_ANG column of the UPWARD file is read
    from row IPHI (the azimuth in degrees) of the table instead of the azimuth's row IP (SOS_ABS_MAIN.F:2466) -- with
    -SOS.View.Dphi > 1 it holds another azimuth's angle, or 0."""
    (nblum, ind_ang, phi, vza, sca_up, i_up, q_up, u_up, ang_up, rate_up, lpol_up,
     sca_dn, i_dn, q_dn, u_dn, ang_dn, rate_dn, lpol_dn) = outputs[:18]
    zal_up, zal_dn = (CTE_TOA_ALT, 0.0) if zout == -1 else (zout, zout)
    up = [main_output_header(itrphi, 1, zal_up, phios)]
    dn = [main_output_header(itrphi, 2, zal_dn, phios)]
    uu, ud = list(up), list(dn)

    def rec55(th, sca, xi, xq, xu, xan, tpol, lpol):
        return ("  %s  %s  %s  %s  %s  %s  %s  %s\n" % (_f7_2(th), _f7_2(sca), fortran_e(xi, 13, 6), fortran_e(xq, 13, 6),
                                                      fortran_e(xu, 13, 6), _f7_2(xan), _f7_2(tpol), fortran_e(lpol, 13, 6))).encode()

    def rec56(ph, th, sca, xi, xq, xu, xan, tpol, lpol):
        return ("  %s  %s  %s %s  %s  %s   %s %s%s\n" % (_f7_2(ph), _f7_2(th), _f7_2(sca), fortran_e(xi, 13, 6), fortran_e(xq, 13, 6),
                                                        fortran_e(xu, 13, 6), _f7_2(xan), _f7_2(tpol), fortran_e(lpol, 13, 6))).encode()

    if itrphi == 1:
        for row, order, sign in ((0, range(nblum - 1, -1, -1), -1.0), (1, range(nblum), 1.0)):
            for j in order:
                a = rec55(sign * vza[j], sca_up[row, j], i_up[row, j], q_up[row, j], u_up[row, j], ang_up[row, j], rate_up[row, j],
                          lpol_up[row, j])
                b = rec55(sign * vza[j], sca_dn[row, j], i_dn[row, j], q_dn[row, j], u_dn[row, j], ang_dn[row, j], rate_dn[row, j],
                          lpol_dn[row, j])
                up.append(a); dn.append(b)
                if ind_ang[j] == 1:
                    uu.append(a); ud.append(b)
    else:
        for ip, iphi in enumerate(range(0, 361, int(pas_phi))):
            for j in range(nblum):
                rest = (i_up[ip, j], q_up[ip, j], u_up[ip, j], ang_up[ip, j], rate_up[ip, j], lpol_up[ip, j])
                up.append(rec56(phi[ip], vza[j], sca_up[iphi, j], *rest))          # row IPHI: the reference's own indexing
                b = rec56(phi[ip], vza[j], sca_dn[ip, j], i_dn[ip, j], q_dn[ip, j], u_dn[ip, j], ang_dn[ip, j], rate_dn[ip, j],
                          lpol_dn[ip, j])
                dn.append(b)
                if ind_ang[j] == 1:
                    uu.append(rec56(phi[ip], vza[j], sca_up[ip, j], *rest))
                    ud.append(b)
    for path, parts in ((path_up, up), (path_down, dn), (user_up, uu), (user_down, ud)):
        if path:
            with open(path, "wb") as f:
                f.write(b"".join(parts))


_HDR_SEP = "#" + "-" * 101 + "\n"
_HDR_COLUMNS = [           # (name, description lines) of the SOS_Up.txt / SOS_Down.txt columns (binding/run_sos.py:255-270)
    ("VZA", ["Viewing Zenith Angle (in degrees)"]),
    ("SCA_ANG", ["Scattering angle (in degrees)"]),
    ("I", ["Stokes parameter I at output altitude z (in sr-1)",
           "normalised to the extraterrestrial solar irradiance (PI * L(z) / Esun)",
           "normalised to the extraterrestrial solar irradiance"]),
    ("Q", ["Stokes parameter Q at output altitude z (in sr-1)", "normalised to the extraterrestrial solar irradiance "]),
    ("U", ["Stokes parameter U at output altitude z (in sr-1)", "normalised to the extraterrestrial solar irradiance "]),
    ("POL_ANG", ["Polarization angle (in degrees). Note: if undefined the value is -999.00"]),
    ("POL_RATE", ["Degree of polarization (in %)"]),
    ("IPOL", ["Polarized intensity at level z (in sr-1)",
              "normalised to the extraterrestrial solar irradiance (PI * Lpol(z) / Esun)"]),
]


def gen_hdr_sos_output(sos_view, updown, zalt):
    """Header of SOS_Up.txt (updown = 1) / SOS_Down.txt (2), the text of binding/run_sos.py:219-278.  One deliberate
    difference: the reference tests `sosView == 1 & updown == 2` (operator precedence makes it always false); here the
    down-looking, fixed-azimuth file gets the 'Viewing direction' wording the reference meant it to have."""
    plane = sos_view == 1
    lines = ["#%s RADIANCE FIELD VERSUS %sVIEWING ZENITH ANGLE\n" % ("UPWARD" if updown == 1 else "DOWNWARD",
                                                                      "THE AZIMUTH ANGLE AND " if plane else ""),
             "# (RELATIVE AZIMUTH AND ALTITUDE ARE FIXED)\n" if plane else "# (ALTITUDE IS FIXED)", _HDR_SEP]
    if plane:
        lines.append("# Relative azimuth (degrees) :\n#\n")
    who = "Viewing direction" if (plane and updown == 2) else "Satellite"
    lines += ["#      Relative azimuth convention :\n",
              "#        180 deg <-> %s and Sun in the same half-plane\n" % who,
              "#          0 deg <-> %s and Sun in opposite half-planes with respect to the zenith direction\n#\n" % who,
              "# Value of the selected altitude for the output (km) : %s\n#\n" % zalt, "# Columns parameters :\n"]
    cols = ([("PHI", ["Relative azimuth Angle (in degrees)"])] if not plane else []) + _HDR_COLUMNS
    for name, desc in cols:
        lines.append("#   %-8s:  %s\n" % (name, desc[0]))
        lines += ["#              %s\n" % d for d in desc[1:]]
    lines.append(_HDR_SEP)
    lines.append("#   %sVZA     SCA_ANG        I              Q              U       POL_ANG  POL_RATE    IPOL\n"
                 % ("" if plane else "PHI      "))
    lines.append("#%s(degrees) (degrees)  (no unit)      (no unit)      (no unit)   (degrees) (pcts)  (no unit)\n"
                 % ("" if plane else "(degrees) "))
    return "".join(lines)


def gen_sos_output(rep_out, sos_view, updown, zalt, nblum, pas_phi, phi, vza, sca_ang, i_out, q_out, u_out,
                   pol_ang_out, pol_rate_out, l_pol_out):
    """run_sos.py:280-317: SOS_Up.txt / SOS_Down.txt (header + data block).  Fixed-azimuth view: the phi + 180 half plane
    first (negated zenith angles, reversed order), then the phi half plane."""
    name = "SOS_Up.txt" if updown == 1 else "SOS_Down.txt"
    with open(os.path.join(rep_out, name), "w") as fic:
        fic.write(gen_hdr_sos_output(sos_view, updown, zalt))
        if sos_view == 1:
            for it in range(nblum - 1, -1, -1):
                fic.write("  %7.2f %7.2f  %13.6e  %13.6e  %13.6e  %7.2f %7.2f %13.6e\n" % (
                    -vza[it], sca_ang[0, it], i_out[0, it], q_out[0, it], u_out[0, it], pol_ang_out[0, it],
                    pol_rate_out[0, it], l_pol_out[0, it]))
            for it in range(nblum):
                fic.write("  %7.2f %7.2f  %13.6e  %13.6e  %13.6e  %7.2f %7.2f %13.6e\n" % (
                    vza[it], sca_ang[1, it], i_out[1, it], q_out[1, it], u_out[1, it], pol_ang_out[1, it],
                    pol_rate_out[1, it], l_pol_out[1, it]))
        else:
            nbphi = math.ceil(360. / pas_phi)
            for ip in range(nbphi):
                for it in range(nblum):
                    fic.write(" %7.2f %7.2f %7.2f  %13.6e  %13.6e  %13.6e  %7.2f %7.2f %13.6e\n" % (
                        phi[ip], vza[it], sca_ang[ip, it], i_out[ip, it], q_out[ip, it], u_out[ip, it],
                        pol_ang_out[ip, it], pol_rate_out[ip, it], l_pol_out[ip, it]))
