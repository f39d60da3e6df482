"""Land surfaces, -SURF.Type 3 (Roujean BRDF), 4 (+ Rondeaux-Herman BPDF), 5 (+ Breon BPDF), 7 (+ Maignan BPDF): SURVEY 8 row f4.
Type 6 (Nadal) is refused by the reference's own SOS_PROC, and here with the same message.

Goldens (make_golden.py proc_land) from the compiled reference: the surface file it generated (every Fourier order, REAL*4),
SOS_Result.bin and the 23 outputs; `cfg5_roujean_maignan` is BASELINE config 5's surface with an LND aerosol, so that all 41
orders of the matrices act inside the solve and the direct terms of SOS_TRPHI (Roujean + Maignan) are exercised in polar view."""
import json
import os

import numpy as np
import pytest

import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAND_CASES = ["land_roujean", "land_rondeaux", "land_breon", "cfg5_roujean_maignan"]


def test_nadal_is_refused_like_the_reference(pkg):
    rs = pkg.run_sos
    kw = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), {
        "-SOS_Main.Wa": 0.67, "-ANG.Thetas": 30.0, "-AP.AbsProfile.Type": 7, "-AER.AOTref": 0.0, "-SURF.Alb": 0.0, "-AP.HR": 8.0,
        "-AP.Psurf": 1013.0, "-SOS.View": 1, "-SOS.View.Phi": 0.0, "-SURF.Type": 6, "-SURF.Ind": 1.5, "-SURF.Roujean.K0": 0.2,
        "-SURF.Roujean.K1": 0.03, "-SURF.Roujean.K2": 0.25, "-SURF.Nadal.Alpha": 0.01, "-SURF.Nadal.Beta": 80.0}))
    with pytest.raises(rs.SosProcError, match="Nadal"):
        rs.sos_proc(**kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name", LAND_CASES)
def test_land_matrices_vs_reference_file(gpu_pkg, name):
    """sosgpu_land_surface against the surface file the reference wrote: REAL*4 elements equal up to last-bit flips (device
    cos / exp differ from the host libm by an ulp before the REAL*4 rounding), same zero pattern, same truncation orders."""
    import torch
    rs, surf = gpu_pkg.run_sos, gpu_pkg.surface
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    nb_lum, nb_mie = int(user["-ANG.Rad.NbGauss"]), int(user["-ANG.Aer.NbGauss"])
    os_nb, os_ns = 2 * nb_mie, 2 * nb_lum
    mu, ga, n0, _ = rs.angles(nb_lum, user["-ANG.Thetas"])
    land = surf.land_model(user["-SURF.Type"], user["-SURF.Roujean.K0"], user["-SURF.Roujean.K1"], user["-SURF.Roujean.K2"],
                           coef_c=user.get("-SURF.Maignan.C", 0.0))
    got = surf.land_matrices(land, mu, ga, user.get("-SURF.Ind", 1.0), os_nb, os_ns, os_nb + os_ns)
    torch.cuda.synchronize()
    got, ref = got.cpu().numpy(), g["rsurf"]
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert err.max() <= 4e-7 * scale, err.max() / scale
    assert np.mean(got != ref) < 2e-2                       # bit-identical but for last-bit flips (wave-order sums, device cos)
    assert np.array_equal(got == 0, ref == 0) or np.mean((got == 0) != (ref == 0)) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", LAND_CASES)
def test_sos_proc_land_vs_reference(gpu_pkg, name, tmp_path):
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    coef = None
    if user["-AER.AOTref"] != 0.0:
        f = str(tmp_path / "Aerosols_user.txt")
        rs.write_aerosols_file(f, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")}, *g["kmat"])
        user["-AER.UserFile"] = f
        coef = 0.0
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    # REAL*4 surface matrices with rare last-bit differences enter linearly: 1e-7 relative on the reflected part
    cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=2e-7)


def test_surface_file_round_trip(pkg, tmp_path):
    rs = pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_land_rondeaux.npz"))
    f = str(tmp_path / "SURF_USER")
    rs.write_surface_file(f, g["rsurf"])
    n = g["rsurf"].shape[-1]
    back = rs.read_surface_file(f, n, g["rsurf"].shape[0] - 1)
    assert back.dtype == np.float32 and np.array_equal(back, g["rsurf"])
    assert os.path.getsize(f) == g["rsurf"].shape[0] * (9 * n * n * 4 + 8)          # one unformatted record per Fourier order
    with pytest.raises(rs.SosProcError):
        rs.read_surface_file(f, n + 1, 3)                                            # made for another angle set
    with pytest.raises(rs.SosProcError):
        rs.read_surface_file(f, n, g["rsurf"].shape[0])                              # one order short


@pytest.mark.gpu
@pytest.mark.parametrize("name", LAND_CASES)
def test_sos_proc_with_the_references_surface_file(gpu_pkg, name, tmp_path):
    """-SURF.File: the reflection matrices come from a user file (here the very file the reference wrote for the case), the
    surface computation is skipped (SOS_PROC.F:3186-3189).  With bit-identical matrices the radiances match at 1e-9."""
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    f = str(tmp_path / "SURF_USER")
    rs.write_surface_file(f, g["rsurf"])
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT", "-SURF.File": f})
    coef = None
    if user["-AER.AOTref"] != 0.0:
        fa = str(tmp_path / "Aerosols_user.txt")
        rs.write_aerosols_file(fa, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")}, *g["kmat"])
        user["-AER.UserFile"] = fa
        coef = 0.0
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    cases.compare_proc_outputs(rs, out, g, coef_tronca=coef)
    user["-SURF.File"] = str(tmp_path / "missing")
    with pytest.raises(rs.SosProcError):
        rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
