"""Randomised parity sweep of the fused solver against the oracle: sizes around every variant boundary (row tiles,
column tiles, field in LDS / in HBM, 4 / 8 waves), all boundary conditions, random optical properties."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _random_case(seed, extreme=False):
    rng = np.random.default_rng(seed + (100003 if extreme else 0))
    S = cases.S
    ng = int(rng.choice([2, 4, 7, 12, 16, 20, 21, 24, 30, 41, 44]))
    nt = int(rng.choice([3, 9, 30, 31, 32, 33, 47, 63, 64, 65, 90, 130]))
    os_nb = int(rng.choice([6, 16, 24, 40]))
    g = float(rng.choice([0.0, 0.4, 0.7, 0.85]))
    if extreme:                                  # thick, strongly forward-peaked, grazing sun, bright ground
        os_nb = int(rng.choice([40, 80]))
        g = float(rng.choice([0.9, 0.95]))
        nt = int(rng.choice([20, 30, 48, 80]))
    mu, w, n0 = S.gauss_angles(ng, float(rng.uniform(75.0, 87.0) if extreme else rng.uniform(5.0, 75.0)))
    al, be, ga, ze = S.hg_phase(os_nb, g)
    nbins = int(rng.integers(1, 4))
    bins = []
    iborm = os_nb
    pure_ray = bool(rng.random() < 0.15)
    a_tr = float(rng.choice([0.0, 0.25]))
    for _ in range(nbins):
        h, x, y, z = S.profile(nt, tau_r=float(rng.uniform(0.02, 0.3)),
                               tau_a=float(rng.uniform(2.0, 8.0) if extreme else rng.uniform(0.05, 1.0)),
                               k_abs=float(np.exp(rng.uniform(np.log(1e-3), np.log(10.0)))))
        if pure_ray:
            x = np.zeros_like(x)
        h, x, y, ib = S.rescale_profile(h, x, y, a_tr, 0.97, 0.95, os_nb)
        iborm = ib
        bins.append((h, x, y, z))
    kw = dict(ro=float(rng.choice([0.6, 0.95]) if extreme else rng.choice([0.0, 0.05, 0.3])))
    surf = rng.random()
    if surf < 0.25:
        kw.update(ifresnel=1, ind_surf=1.34)
    elif surf < 0.45:
        kw.update(imat_surf=1)
    if rng.random() < 0.3:
        kw["zout"] = float(rng.uniform(0.2, 8.0))
    if rng.random() < 0.15:
        kw["ipolar"] = 0
    if rng.random() < 0.1:
        kw["igmax"] = 4
    if kw.get("imat_surf"):
        kw["rsurf"] = cases._surf_matrices(len(mu), iborm, seed)
    return dict(name="fuzz%d" % seed, rmu=mu, ga=w, n0=n0, os_nb=os_nb, coefs=(al, be, ga, ze), bins=bins, iborm=iborm,
                kw=kw)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SOS_FUZZ_N", "24"))))     # SOS_FUZZ_N=300 for a long hunt
def test_random_configuration(gpu_pkg, oracle, seed):
    case = _random_case(seed)
    got = cases.run_gpu(gpu_pkg, case)
    for b, g in enumerate(got):
        ref = cases.run_cpu(oracle, case, b)
        what = "%s N=%d NT=%d bin %d %s" % (case["name"], len(case["rmu"]), len(case["bins"][b][0]) - 1, b,
                                           {k: v for k, v in case["kw"].items() if k != "rsurf"})
        assert len(g["records"]) == len(ref["records"]), what
        assert np.array_equal(g["ig_counts"], ref["ig_counts"]), what
        cases.compare_records(g["records"], ref["records"], 1e-9, what)
        assert abs(g["emoins"] - ref["emoins"]) <= 1e-9 * abs(ref["emoins"]) + 1e-300, what
        assert abs(g["eplus"] - ref["eplus"]) <= 1e-9 * abs(ref["eplus"]) + 1e-300, what


@pytest.mark.parametrize("seed", range(int(os.environ.get("SOS_FUZZ_EXTREME_N", "6"))))
def test_extreme_configuration(gpu_pkg, oracle, seed):
    """Optically thick aerosol (tau 2..8), g = 0.9 / 0.95, sun at 75..87 degrees, bright ground: many scattering orders and
    Fourier orders (the stop tests work hardest here)."""
    case = _random_case(seed, extreme=True)
    got = cases.run_gpu(gpu_pkg, case)
    for b, g in enumerate(got):
        ref = cases.run_cpu(oracle, case, b)
        what = "%s(extreme) N=%d NT=%d bin %d" % (case["name"], len(case["rmu"]), len(case["bins"][b][0]) - 1, b)
        assert len(g["records"]) == len(ref["records"]), what
        assert np.array_equal(g["ig_counts"], ref["ig_counts"]), what
        cases.compare_records(g["records"], ref["records"], 1e-9, what)
