"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Tolerance: 1e-9 relative on every Stokes Fourier record (north star), identical Fourier-order and
scattering-order counts (data-dependent stops must take the reference's decisions)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_sos_os_vs_oracle(gpu_pkg, oracle, name):
    case = cases.make_case(name)
    got = cases.run_gpu(gpu_pkg, case)
    for b, g in enumerate(got):
        ref = cases.run_cpu(oracle, case, b)
        assert len(g["records"]) == len(ref["records"]), (name, b, len(g["records"]), len(ref["records"]))
        assert np.array_equal(g["ig_counts"], ref["ig_counts"]), (name, b, g["ig_counts"], ref["ig_counts"])
        worst = cases.compare_records(g["records"], ref["records"], 1e-9, "%s bin %d" % (name, b))
        assert abs(g["emoins"] - ref["emoins"]) <= 1e-9 * abs(ref["emoins"]) + 1e-300
        assert abs(g["eplus"] - ref["eplus"]) <= 1e-9 * abs(ref["eplus"]) + 1e-300
        print(name, b, "F", len(ref["records"]), "worst rel", worst)


@pytest.mark.parametrize("is_", [0, 1, 2, 3, 40, 80])
def test_noyaux_vs_oracle(gpu_pkg, oracle, is_):
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(40, 35.0)
    al, be, ga, ze = S.hg_phase(80, 0.75)
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze)
    got = cx.noyaux_fetch(is_)
    ref = oracle.noyaux(is_, -mu[n0 - 1], mu, 80, al, be, ga, ze)
    for k in ["BP", "GR", "GT", "ARR", "ART", "ATT", "XPL", "XRL", "XTL"]:
        scale = np.abs(ref[k]).max() + 1e-300
        assert np.abs(got[k] - ref[k]).max() <= 1e-13 * scale, (is_, k, np.abs(got[k] - ref[k]).max() / scale)
    cx.close()
