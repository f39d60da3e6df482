"""Aerosol models: Mie theory + size distribution + truncated Legendre expansion, against the Aerosols.txt
content the compiled reference produced for the same parameters (the aer_* arrays of tests/golden/sos_proc_*.npz, captured by
make_golden.py proc_aer / proc_land / proc_ckd) -- mono-modal LND (configs 2, 3, 5), bimodal LND (config 4).
The reference itself passes the Mie results through a REAL*4 file, so agreement is at the REAL*4 level: coefficients to 2e-6 of
beta_0 = 1; the printed truncation coefficient and albedo (F9.5) exactly or within their last digit."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["cfg2_lnd_lambert", "cfg4_glitter_bilnd", "cfg5_roujean_maignan", "ckd_h2o_o2_25bins_flatsea"]


def test_alpha_grid_and_angles(pkg):
    A = pkg.aerosols
    g = A.alpha_grid(0.0001, 100.0)
    assert g[0] == 0.0001 and g[-1] <= 100.0 and np.all(np.diff(g) > 0)
    # 1e-4 steps up to 0.1, 1e-3 to 1, 1e-2 to 10, 0.05 to 30, 0.1 to 100 (SOS_MIE.F:437-443): about 3900 records
    assert 3800 < len(g) < 4000 and abs(np.diff(g)[0] - 1e-4) < 1e-11 and abs(np.diff(g)[-1] - 0.1) < 1e-8
    xmu, xhr = A.mie_angles(40)
    assert len(xmu) == 81 and xmu[40] == 0 and np.allclose(xmu[41:], -xmu[:40][::-1]) and abs(xhr[41:].sum() - 1.0) < 1e-13
    assert abs(xmu[41] - 0.19511383256794e-01) < 1e-15          # first row of the reference's Aer_UsedAngles.txt


def test_decompo_legendre_recovers_a_known_expansion(pkg):
    """A phase matrix synthesised from known coefficients on the Gauss nodes comes back through the expansion (no
    truncation): the quadrature is exact for polynomials of degree < 4 N."""
    A = pkg.aerosols
    xmu, xhr = A.mie_angles(24)
    nb = 20
    l = np.arange(nb + 1)
    beta = (2 * l + 1) * 0.6 ** l
    P = np.polynomial.legendre.legval(xmu, beta)
    z = np.zeros_like(P)
    d = A.decompo_legendre(0, xmu, xhr, nb, P, z, P.copy(), z)
    assert np.allclose(d["beta"], beta / beta[0], rtol=0, atol=1e-12) and d["coef_tronca"] == 0.0 and d["itronc"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_aerosol_model_vs_reference_aerosols_file(gpu_pkg, name):
    rs, A = gpu_pkg.run_sos, gpu_pkg.aerosols
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    nb_mie = int(user["-ANG.Aer.NbGauss"])
    got = A.aerosols(p, user["-SOS_Main.Wa"], user["-AER.AOTref"], nb_mie, 2 * nb_mie, at_waref=False)
    for k in ("alpha", "beta", "gamma", "zeta"):
        ref = g["aer_" + k]
        assert np.abs(got[k] - ref).max() <= 2e-6, (k, np.abs(got[k] - ref).max())
    assert abs(got["a_tronc"] - float(g["aer_a_tronc"])) <= 1.001e-5 and abs(got["piztr"] - float(g["aer_piztr"])) <= 1.001e-5
    assert np.allclose([got["kmat1"], got["kmat2"]], g["kmat"], rtol=6e-5)          # printed E13.5: five digits
    if "coef_tronca" in g.files and float(g["coef_tronca"]) != 0.0:
        assert abs(got["coef_tronca"] - float(g["coef_tronca"])) <= 2e-6


@pytest.mark.gpu
def test_sos_proc_with_its_own_aerosol_model(gpu_pkg):
    """-AER.Model 0 end to end (BASELINE config 2 with the LND model): radiances within 2e-6 of the reference run (the
    difference is the REAL*4 Mie hand-off, not the solver: the same case is pinned at 1e-9 through -AER.UserFile)."""
    import cases
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_cfg2_lnd_lambert.npz"))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    cases.compare_proc_outputs(rs, out, g, rtol=5e-6)
