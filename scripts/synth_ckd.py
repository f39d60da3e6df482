"""Synthetic CKD coefficient tables in the reference's file format (fic/COEFF_CKD/10cmm1/coef_<GAS>_<numax>_<numin>_10cmm1,
SOS_SUB_TRS.F:481-907; layout documented in each file's header) for the hyperspectral benchmark: the GPU box has no reference
checkout and the checkout here lacks half of the H2O files.  The NUMBER of exponential terms of every gas and interval is the
reference's own (scripts/data/nexp_10cmm1.txt: mean 2.3 CKD bins per interval, max 125, 5736 bins over the 2500 intervals);
the coefficient VALUES are synthetic: a gas with n terms has column optical depths log-uniform in [1e-3, 30],
seeded, strongest term first; weights Dirichlet(1).  Writes only the files of the requested wavenumber range."""
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GASES = ["H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "NO2"]
COLUMN = [5e22, 8.5e21, 9e18, 6.6e18, 2e18, 3.6e19, 4.5e24, 5e15]          # molecules / cm2, rough total columns
T_GRID = [160., 180., 200., 220., 240., 260., 280., 300., 320.]
P_GRID = [7.0e-3, 1.1e-2, 1.7e-2, 2.8e-2, 4.4e-2, 6.9e-2, 0.11, 0.174, 0.276, 0.438, 0.694, 1.1, 1.743, 2.763, 4.379, 6.941, 11.0,
          17.434, 27.631, 43.792, 69.405, 110.0, 174.338, 219.479, 276.308, 347.851, 437.918, 551.306, 694.053, 873.761, 1098.9]
C_GRID = [1.6094e-07, 5.0843e-07, 1.6078e-06, 5.0843e-06, 1.6078e-05, 5.0843e-05, 1.6078e-04, 5.0843e-04, 1.6078e-03, 5.0843e-03,
          1.6078e-02, 5.0792e-02]


def nexp_table():
    rows = [ln.strip() for ln in open(os.path.join(HERE, "data", "nexp_10cmm1.txt")) if not ln.startswith("#")]
    return np.array([[int(c) for c in r] for r in rows], dtype=np.int32)       # [8][2500]


def write_tables(root, nu_lo=2500.0, nu_hi=27500.0, seed=2024, so2_no2=None):
    """Create root/fic/COEFF_CKD/10cmm1/... for the 500 cm-1 files overlapping [nu_lo, nu_hi] and root/fic/SO2-NO2 (copied from
    `so2_no2`, default tests/golden/fic/SO2-NO2).  Returns the number of files written."""
    nexp = nexp_table()
    d = os.path.join(root, "fic", "COEFF_CKD", "10cmm1")
    os.makedirs(d, exist_ok=True)
    shutil.copy(so2_no2 or os.path.join(os.path.dirname(HERE), "tests", "golden", "fic", "SO2-NO2"), os.path.join(root, "fic", "SO2-NO2"))
    nfiles = 0
    for f in range(50):
        numax, numin = 27500 - 500 * f, 27000 - 500 * f
        if numax <= nu_lo or numin >= nu_hi:
            continue
        for k, gas in enumerate(GASES):
            rng = np.random.default_rng(seed + 1000 * k + f)
            head = 21 if k == 0 else 18
            out = ["synthetic CKD table (scripts/synth_ckd.py)"] * head
            out.append("%d %d 10" % (numax, numin))
            out.append("%d" % len(T_GRID)); out.append(" ".join("%8.2f" % t for t in T_GRID))
            out.append("%d" % len(P_GRID)); out.append(" ".join("%9.3f" % p for p in P_GRID))
            nc = 1
            if k == 0:
                nc = len(C_GRID)
                out.append("%d" % nc); out.append(" ".join("%.4e" % c for c in C_GRID))
            for i in range(50):
                hi, lo = numax - 10.0 * i, numax - 10.0 * (i + 1)
                n = int(nexp[k, 50 * f + i])
                if n == 1 and rng.uniform() < 0.7:
                    n = 0                                   # a single term is mostly "no absorption here" (NMAXAI = 0)
                out.append("%.8f %.8f %.8f %.1f %.1f  %d" % (1e4 / hi, 2e4 / (hi + lo), 1e4 / lo, hi, lo, n))
                if n == 0:
                    continue
                ai = rng.dirichlet(np.ones(n))
                out.append(" ".join("%.6e" % a for a in ai))
                tau = np.sort(np.exp(rng.uniform(np.log(1e-3), np.log(30.0), n)))[::-1] if n > 1 else np.array([10 ** rng.uniform(-3, -1)])
                for t in range(n):
                    kbase = tau[t] / COLUMN[k]
                    for ic in range(nc):
                        for ip, p in enumerate(P_GRID):
                            kk = kbase * (0.6 + 0.4 * (p / 1013.0) ** 0.5) * (1.0 + 0.05 * ic)
                            vals = " ".join("%.5e" % (kk * (1.0 + 0.1 * (tt - 260.) / 100.)) for tt in T_GRID)
                            out.append(("%d %d %d " % (t + 1, ic + 1, ip + 1) if k == 0 else "%d %d " % (t + 1, ip + 1)) + vals)
            with open(os.path.join(d, "coef_%s_%d_%d_10cmm1" % (gas, numax, numin)), "w") as g:
                g.write("\n".join(out) + "\n")
            nfiles += 1
    return nfiles


if __name__ == "__main__":
    import sys
    print(write_tables(sys.argv[1]), "files")
