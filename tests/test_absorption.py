"""Gas absorption by the CKD method.

Fixtures, all made from the compiled reference with its full CKD tables (tests/golden/make_golden.py absorption / proc_ckd):
  absorption.npz                layer amounts RO, profile table, LAMB1, NEXP, KDIS_AI and TAUABS(50) of every bin, from
                                sos_prepa_absprofile_ / sos_absprofile_
  sos_proc_ckd_*.npz            end-to-end SOS_PROC of multi-bin bands: O2-A (5 bins), H2O x O2 at 15925 cm-1 (25 bins, LND
                                aerosol, flat sea: BASELINE config 3's shape), -SOS.AbsModeCKD 2
  fic/                          trimmed copies of the CKD data files the cases read (make_fic_fixture.py): $SOS_ABS_ROOT here
CPU: the host steps (absorption.py) against absorption.npz; GPU: sosgpu_absprofile and run_sos.sos_proc end to end."""
import json
import os

import numpy as np
import pytest

import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ABS_CASES = ["o2a_mls", "h2o_o2_trop_user", "h2o_o2_subarctic", "o2a_us62_nopsurf"]
CKD_CASES = ["ckd_o2a_5bins", "ckd_h2o_o2_25bins_flatsea", "ckd_o2a_mode2", "cfg5_ckd_maignan_25bins", "ckd_userprofile_25bins"]


@pytest.fixture()
def fic(monkeypatch):
    monkeypatch.setenv("SOS_ABS_ROOT", GOLD)
    return GOLD


def _prep(A, g, name):
    wa, nustep, psurf, h2o, o3, co2, ch4, typ = g[name + "_args"]
    return A.prepa_absprofile(wa, nustep, psurf, h2o, o3, co2, ch4, int(typ))


@pytest.mark.parametrize("name", ABS_CASES)
def test_prepa_and_absprofile_vs_reference(pkg, fic, name):
    """SOS_PREPA_ABSPROFILE bit for bit (RO, profile table, LAMB1, NEXP, KDIS_AI), bins in the reference's loop order,
    TAUABS of every bin to 1e-13 of the column (glibc vs flang exp / log)."""
    A = pkg.absorption
    g = np.load(os.path.join(GOLD, "absorption.npz"))
    p = _prep(A, g, name)
    assert p["lamb1"] == int(g[name + "_lamb1"]) and p["nu"] == float(g[name + "_nu"])
    assert np.array_equal(p["altabs"], g[name + "_altabs"])
    assert np.array_equal(p["userprofil"], g[name + "_userprofil"])
    assert np.array_equal(p["ro"], g[name + "_ro"])
    assert np.array_equal(p["nexp"], g[name + "_nexp"]) and np.array_equal(p["kdis_ai"], g[name + "_kdis_ai"])
    ik, aik, s = A.bins(p)
    assert np.array_equal(ik, g[name + "_ik"]) and abs(aik.sum() - 1.0) < 1e-14
    xk, ro = A.layer_tables(p)
    ref = g[name + "_tau"]
    for b in range(len(ik)):
        tau = A.absprofile_host(xk, ro, ik[b])
        assert np.all(np.abs(tau - ref[b]) <= 2e-15 + 1e-13 * ref[b, -1]), (name, b)


def test_bin_order_and_weights_are_the_reference_loop_nest(pkg):
    """a17: gas 1 outermost ... gas 8 innermost, AIK = product left to right / serial sum (SOS_PROC.F:3381-3404,3481-3487),
    written out as the literal eight nested loops."""
    g = np.load(os.path.join(GOLD, "absorption.npz"))
    nexp, a = g["h2o_o2_trop_user_nexp"], g["h2o_o2_trop_user_kdis_ai"]
    ik, aik, s = pkg.ckd.ckd_bin_weights(nexp, a)
    exp_ik, exp_w = [], []
    for i1 in range(nexp[0]):
        for i2 in range(nexp[1]):
            for i3 in range(nexp[2]):
                for i4 in range(nexp[3]):
                    for i5 in range(nexp[4]):
                        for i6 in range(nexp[5]):
                            for i7 in range(nexp[6]):
                                for i8 in range(nexp[7]):
                                    w = a[i1, 0] * a[i2, 1] * a[i3, 2] * a[i4, 3] * a[i5, 4] * a[i6, 5] * a[i7, 6] * a[i8, 7]
                                    exp_ik.append([i1 + 1, i2 + 1, i3 + 1, i4 + 1, i5 + 1, i6 + 1, i7 + 1, i8 + 1])
                                    exp_w.append(w)
    tot = 0.0
    for w in exp_w:
        tot = tot + w
    assert len(aik) == 25 and np.array_equal(ik, np.array(exp_ik)) and s == tot
    assert np.array_equal(aik, np.array(exp_w) / tot)


@pytest.mark.skipif(not os.path.isdir("/root/reference/fic"), reason="full CKD tables only exist in the authoring container")
def test_trimmed_tables_equal_full(pkg, monkeypatch):
    A = pkg.absorption
    g = np.load(os.path.join(GOLD, "absorption.npz"))
    for name in ABS_CASES:
        monkeypatch.setenv("SOS_ABS_ROOT", GOLD)
        a = _prep(A, g, name)
        monkeypatch.setenv("SOS_ABS_ROOT", "/root/reference")
        b = _prep(A, g, name)
        assert np.array_equal(a["nexp"], b["nexp"]) and np.array_equal(a["kdis_ai"], b["kdis_ai"])
        for k in range(8):
            assert np.array_equal(a["ki"][k], b["ki"][k]), (name, k)


def test_errors_like_the_reference(pkg, fic, monkeypatch):
    A = pkg.absorption
    with pytest.raises(A.AbsorptionError):          # resolution outside {1, 5, 10} (READ_CKD_COEFF ERROR_905)
        A.prepa_absprofile(0.762, 2.0, 1013.0, -999., -999., -999., -999., 2)
    with pytest.raises(A.AbsorptionError):          # wavelength outside the CKD range (ERROR_905 of SOS_PREPA_ABSPROFILE)
        A.prepa_absprofile(0.30, 10.0, 1013.0, -999., -999., -999., -999., 2)
    with pytest.raises(A.AbsorptionError):          # table not shipped (H2O tables are mostly missing upstream as well)
        A.prepa_absprofile(0.910, 10.0, 1013.0, -999., -999., -999., -999., 2)
    monkeypatch.delenv("SOS_ABS_ROOT")
    with pytest.raises(A.AbsorptionError):
        A.prepa_absprofile(0.762, 10.0, 1013.0, -999., -999., -999., -999., 2)


@pytest.mark.gpu
def test_absprofile_kernel_vs_host(gpu_pkg, fic):
    """sosgpu_absprofile (all bins in one launch) == the per-bin loop of SOS_ABSPROFILE, and == the reference's TAUABS."""
    import torch
    A = gpu_pkg.absorption
    g = np.load(os.path.join(GOLD, "absorption.npz"))
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(8, 35.0)
    al, be, ga, ze = S.hg_phase(16, 0.5)
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=16)
    for name in ABS_CASES:
        p = _prep(A, g, name)
        ik, aik, _ = A.bins(p)
        xk, ro = A.layer_tables(p)
        tabs = cx.absorption_profiles(ik, xk, ro)
        torch.cuda.synchronize()
        tabs = tabs.cpu().numpy()
        ref = g[name + "_tau"]
        for b in range(len(ik)):
            host = A.absprofile_host(xk, ro, ik[b])
            assert np.all(np.abs(tabs[b] - host) <= 2e-15 + 1e-14 * host[-1]), (name, b)       # -ln(TRS) near TRS = 1
            assert np.all(np.abs(tabs[b] - ref[b]) <= 2e-15 + 1e-13 * ref[b, -1]), (name, b)
    cx.close()


def _floats(text):
    out = []
    for tok in text.replace("=", " ").replace(":", " ").split():
        try:
            out.append(float(tok))
        except ValueError:
            pass
    return np.array(out)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CKD_CASES)
def test_sos_proc_ckd_vs_reference(gpu_pkg, fic, name, tmp_path):
    """Multi-bin CKD bands through the drop-in: bin weights -> absorption profiles -> SOS_PROFILE (all bins on the device)
    -> SOS_OS -> SOS_AGGREGATE -> SOS_TRPHI, against the reference's 23 outputs and its SOS_Result.bin; the -SOS.Trans and
    -SOS.Flux files number for number."""
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v)
            for k, v in json.loads(str(g["user_json"])).items()}
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS_Main.ResRoot": str(tmp_path)})
    user.setdefault("-SOS.Flux", "NO_OUTPUT")
    coef = None
    if user["-AER.AOTref"] != 0.0:
        f = str(tmp_path / "Aerosols_user.txt")
        rs.write_aerosols_file(f, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")}, *g["kmat"])
        user["-AER.UserFile"] = f
        coef = 0.0
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    # land surfaces: REAL*4 reflection matrices with rare last-bit differences enter linearly (cf. tests/test_land.py)
    rtol = 2e-7 if int(user.get("-SURF.Type", 0)) >= 3 else 1e-9
    cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=rtol)
    # SOS_Result.bin: the reference appends one all-zero record per aggregated bin after the first (SOS_AGGREGATE.F:357-413)
    from oracle import ref_ctypes as R
    got = np.array(R.read_fortran_records(str(tmp_path / "SOS" / "SOS_Result.bin")))
    ref = g["result_bin"]
    assert len(got) <= len(ref) and np.all(ref[len(got):] == 0.0)
    n = int(g["nblum"]); w = 2 * n + 1
    for r in (got, ref):
        r[:, [n, w + n, 2 * w + n]] = 0.0             # slot jj = 0 is never initialised by the reference
    scale = np.abs(ref[:, 2 * w:]).max()
    assert np.all(np.abs(got - ref[:len(got)]) <= rtol * np.abs(ref[:len(got)]) + max(1e-12, 1e-3 * rtol) * scale)
    for key in ("file_trans", "file_flux"):
        if key in g.files:
            fname = user["-SOS.Trans"] if key == "file_trans" else user["-SOS.Flux"]
            mine, theirs = _floats(open(str(tmp_path / "SOS" / fname)).read()), _floats(str(g[key]))
            assert mine.shape == theirs.shape, key
            assert np.all(np.abs(mine - theirs) <= 1.001e-4 + 1e-9 * np.abs(theirs)), (key, np.abs(mine - theirs).max())


def test_vectorised_layer_tables_equal_the_scalar_restatement(pkg, fic):
    """layer_tables interpolates the 49 layers of all (term, gas) pairs together on brackets kept per atmosphere; it must
    reproduce, bit for bit, the layer-by-layer form that follows COEFF_ABS_CKD statement for statement (clamps carried from gas
    to gas included), and so must the pair-by-pair form between the two."""
    A = pkg.absorption
    for wa, ap, h2o in ((0.762, 2, -999.), (1.0e4 / 15925.0, 1, 2.5), (1.0e4 / 15925.0, 6, -999.)):
        prep = A.prepa_absprofile(wa, 10.0, 1013.0, h2o, -999., -999., -999., ap)
        xv, rv = A.layer_tables(prep)
        xs, rs_ = A.layer_tables_scalar(prep)
        assert np.array_equal(xv, xs) and np.array_equal(rv, rs_) and (xs != 0).sum() > 100
        xp, rp = A.layer_tables_by_pair(prep)
        assert np.array_equal(xp, xs) and np.array_equal(rp, rs_)
    # the wavelengths of a spectrum in one pass (sos_spectrum): the same tables, bit for bit, and served through the prefetch
    reqs = [(0.762, 10.0, 1013.0, -999., -999., -999., -999., 2, None), (1.0e4 / 15925.0, 10.0, 1013.0, 2.5, -999., -999., -999., 1, None),
            (1.0e4 / 15925.0, 10.0, 1013.0, -999., -999., -999., -999., 6, None), (0.7625, 10.0, 1013.0, -999., -999., -999., -999., 2, None)]
    one = [A.layer_tables(A.prepa_absprofile(*r)) for r in reqs]
    try:
        assert A.prefetch_gas_tables(reqs + reqs[:1]) == len(reqs)
        for r, (x1, r1) in zip(reqs, one):
            prep = A.prepa_absprofile(*r)
            assert "_layer_tables" in prep
            xm, rm = A.layer_tables(prep)
            assert np.array_equal(xm, x1) and np.array_equal(rm, r1)
    finally:
        A.drop_prefetched_gas_tables()
    assert "_layer_tables" not in A.prepa_absprofile(*reqs[0])
