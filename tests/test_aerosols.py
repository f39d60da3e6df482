"""Aerosol models: Mie theory + size distribution + truncated Legendre expansion, against the Aerosols.txt
content the compiled reference produced for the same parameters (the aer_* arrays of tests/golden/sos_proc_*.npz, captured by
make_golden.py proc_aer / proc_land / proc_ckd / aer_models) -- mono-modal LND (configs 2, 3, 5), bimodal LND (config 4),
WMO, Shettle & Fenn, external phase functions, user mixtures.
Pinned at full precision (round 3): the device Mie kernel reproduces the reference's REAL*4 MIE-file records BIT FOR BIT
(tests/golden/mie_chain.npz: SOS_MIE records of five size-parameter ranges and of a whole run), the host chain SOS_GRANU ->
SOS_DECOMPO_LEGENDRE fed the reference's own records reproduces its Aerosols.txt byte for byte, and together they give the
Aerosols.txt coefficients of every model digit for digit (a handful of entries one unit of the eighth digit off: a 1e-16
difference of the libm exp/log sitting on an E15.8 rounding boundary) and the radiances at the 1e-9 bar
(profiles/r03_mie_parity.txt).  Round 2's 2e-6 / 2e-5 tolerances came from pairwise numpy sums in the size-distribution
integral where the Fortran loop adds record by record."""
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
    # (1e-10: the nodes and weights are the reference's own -- SOS_GAUSS stops its Newton iteration at 1e-15 in the node and
    # its weights carry 1e-14 .. 1e-13 of that; numpy's leggauss would give 1e-12 here)
    assert np.allclose(d["beta"], beta / beta[0], rtol=0, atol=1e-10) and d["coef_tronca"] == 0.0 and d["itronc"] == 0


def _coefficients_digit_for_digit(got, g):
    """alpha, beta, gamma, zeta as Aerosols.txt prints them (E15.8) against the reference's file: equal digit for digit, up to
    one unit of the eighth significant digit (E15.8 rounding boundaries) or -- for the small high-order coefficients, which are
    differences of O(1) sums -- 2e-12 of the largest coefficient: the device exp / log of the size distribution differ from the
    host libm's by an ulp.  At most 5 % of the entries differ at all."""
    nd = ntot = 0
    scale = max(float(np.abs(g["aer_" + k]).max()) for k in ("alpha", "beta", "gamma", "zeta"))
    for k in ("alpha", "beta", "gamma", "zeta"):
        ref = g["aer_" + k]
        assert np.all(np.abs(got[k] - ref) <= 1.001e-7 * np.abs(ref) + 2e-12 * scale), (k, np.abs(got[k] - ref).max())
        nd += int(np.sum(got[k] != ref))
        ntot += len(ref)
    assert nd <= 0.05 * ntot, (nd, ntot)


def test_host_chain_on_the_references_own_mie_records(pkg, monkeypatch, tmp_path):
    """SOS_GRANU (numpy restatement, tests/aerosol_loops.py) + the product's SOS_DECOMPO_LEGENDRE + Aerosols.txt writer, fed
    the records of the MIE file the reference's run left behind (mie_chain.npz): the reference's Aerosols.txt byte for byte.
    Isolates the host chain from the device kernels; k_granu is checked against granu_host on the GPU below."""
    A, rs = pkg.aerosols, pkg.run_sos
    g = np.load(os.path.join(GOLD, "mie_chain.npz"))
    user = json.loads(str(g["user_json"]))
    rec = {k[4:]: g[k] for k in g.files if k.startswith("mie_") and k != "mie_file_name"}
    rec["alphaf"] = float(rec["alphaf"])
    import aerosol_loops
    calls = []

    def integral_of_the_reference_records(xmu, rn, in_, alphaf, igranu, v1, v2, v3, wa, device=0):
        calls.append((rn, in_, alphaf, len(xmu)))
        return aerosol_loops.granu_host(rec, igranu, v1, v2, v3, wa)

    monkeypatch.setattr(A, "size_integral", integral_of_the_reference_records)
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    got = A.aerosols(p, user["-SOS_Main.Wa"], user["-AER.AOTref"], 12, 24, at_waref=True)
    # the grid the product asks for is the one the reference's file was written for (its name carries index and range)
    assert calls == [(float(rec["rn"]), float(rec["in_"]), rec["alphaf"], 25)]
    assert str(g["mie_file_name"]) == "MIE1.450-0.00300-0.0001-00100.00-MU12"
    assert np.array_equal(A.alpha_grid(A.MIE_ALPHAMIN, rec["alphaf"]).astype(np.float32), rec["alpha"])
    f = str(tmp_path / "Aerosols.txt")
    rs.write_aerosols_file(f, got, got["kmat1"], got["kmat2"])
    assert open(f).read() == str(g["aerosols_txt"])


@pytest.mark.gpu
def test_mie_kernel_records_vs_the_references_mie_file(gpu_pkg):
    """k_mie against SOS_MIE element by element, REAL*4 as the MIE file holds them: five size-parameter ranges (0.0001..0.02,
    0.9..12, 95..130, the LDS / HBM-scratch boundary 840..860, the WMO dust-like end 3980..4000) and the 3900 records of a whole
    run.  Measured: every entry identical (profiles/r03_mie_parity.txt); the test allows 0.1 % last-bit flips (device libm)."""
    A = gpu_pkg.aerosols
    g = np.load(os.path.join(GOLD, "mie_chain.npz"))

    def same(got, ref, tag):
        for k in ("alpha", "qext", "qsca", "imie", "qmie", "umie"):
            a = np.ascontiguousarray(got[k], np.float32).view(np.int32).astype(np.int64)
            b = np.ascontiguousarray(ref[k], np.float32).view(np.int32).astype(np.int64)
            assert np.abs(a - b).max() <= 1 and np.mean(a != b) <= 1e-3, (tag, k, np.abs(a - b).max(), np.mean(a != b))
        assert np.allclose(got["g"], ref["g"], rtol=1e-13, atol=0), tag

    xmu = g["range_xmu"]
    for name in ("small", "mid", "large", "lds_edge", "dustlike"):
        rn, in_, a0, a1 = g["range_" + name]
        ref = {k: g["range_%s_%s" % (name, k)] for k in ("alpha", "qext", "qsca", "g", "imie", "qmie", "umie")}
        same(A.mie_records(xmu, rn, in_, a0, a1), ref, name)
    ref = {k[4:]: g[k] for k in g.files if k.startswith("mie_") and k != "mie_file_name"}
    xm, _ = A.mie_angles(int(ref["nbmu"]))
    same(A.mie_records(xm, float(ref["rn"]), float(ref["in_"]), A.MIE_ALPHAMIN, float(ref["alphaf"])), ref, "chain")


@pytest.mark.gpu
def test_device_size_integral_vs_host_restatement(gpu_pkg):
    """k_granu (record-order sums on the device) against the numpy restatement on the same records: log-normal and Junge laws,
    cross sections and phase functions to 1e-14 (device exp / log / pow against glibc's), same record count used."""
    import aerosol_loops
    A = gpu_pkg.aerosols
    xmu, _ = A.mie_angles(12)
    for igranu, v1, v2, v3, wa, af in ((1, 0.12, 0.45, -999.0, 0.865, 100.0), (1, 0.8, 0.6, -999.0, 0.55, 300.0),
                                       (2, 0.05, 4.2, 12.0, 0.865, 100.0), (2, 0.1, 3.5, 50.0, 1.6, 200.0)):
        rec = A.mie_records(xmu, 1.45, -0.003, A.MIE_ALPHAMIN, af)
        ref = aerosol_loops.granu_host(rec, igranu, v1, v2, v3, wa)
        got = A.size_integral(xmu, 1.45, -0.003, af, igranu, v1, v2, v3, wa)
        for a, b in zip(got[:3], ref[:3]):
            assert abs(a - b) <= 1e-13 * abs(b), (igranu, a, b)
        for a, b in zip(got[3:], ref[3:]):
            assert np.allclose(a, b, rtol=1e-13, atol=1e-15 * np.abs(b).max()), igranu


@pytest.mark.gpu
def test_size_integrals_queued_ahead_equal_the_direct_calls(gpu_pkg):
    """sosgpu_granu_batch (what sos_spectrum queues for a batch of wavelengths) against sosgpu_granu, bit for bit: 70 jobs over
    three launches, two record sets of different length, both laws; then the collect -> prefetch -> model pass of a bimodal
    aerosol model against the plain pass."""
    import torch
    A, rs = gpu_pkg.aerosols, gpu_pkg.run_sos
    xmu, _ = A.mie_angles(12)
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(70):
        if k % 3 == 2:
            jobs.append((1.33, -0.001, 100.0, 2, 0.03 + 0.01 * rng.random(), 3.5 + rng.random(), 8.0 + 4 * rng.random(),
                         0.4 + 1.5 * rng.random()))
        else:
            jobs.append((1.45, -0.003, 300.0 if k % 2 else 100.0, 1, 0.05 + 0.5 * rng.random(), 0.3 + 0.4 * rng.random(), -999.0,
                         0.4 + 1.5 * rng.random()))
    direct = [A.size_integral(xmu, rn, in_, af, ig, v1, v2, v3, wa) for rn, in_, af, ig, v1, v2, v3, wa in jobs]
    keys = [A._granu_key(xmu, rn, in_, af, ig, v1, v2, v3, wa, 0) for rn, in_, af, ig, v1, v2, v3, wa in jobs]
    try:
        assert A.prefetch_size_integrals(keys + keys[:5]) == 70
        ahead = [A.size_integral(xmu, rn, in_, af, ig, v1, v2, v3, wa) for rn, in_, af, ig, v1, v2, v3, wa in jobs]
    finally:
        A.drop_prefetched_size_integrals()
    for d, a in zip(direct, ahead):
        assert d[:3] == a[:3]
        for x, y in zip(d[3:], a[3:]):
            assert np.array_equal(x, y)
    # the model code in collect mode names exactly the integrals its real pass asks for
    g = np.load(os.path.join(GOLD, "sos_proc_cfg4_glitter_bilnd.npz"))
    user = json.loads(str(g["user_json"]))
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    nb_mie = int(user["-ANG.Aer.NbGauss"])
    plain = A.aerosols(p, 0.7, 0.1, nb_mie, 2 * nb_mie, at_waref=False)
    with A.collect_size_integrals() as reqs:
        assert A.aerosols(p, 0.7, 0.1, nb_mie, 2 * nb_mie, at_waref=False) is None
    assert len(reqs) >= 2
    try:
        A.prefetch_size_integrals(reqs)
        ahead = A.aerosols(p, 0.7, 0.1, nb_mie, 2 * nb_mie, at_waref=False)
    finally:
        A.drop_prefetched_size_integrals()
    for k in ("alpha", "beta", "gamma", "zeta"):
        assert np.array_equal(plain[k], ahead[k]), k
    assert plain["kmat1"] == ahead["kmat1"] and plain["coef_tronca"] == ahead["coef_tronca"]
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_aerosol_model_vs_reference_aerosols_file(gpu_pkg, name):
    rs, A = gpu_pkg.run_sos, gpu_pkg.aerosols
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    nb_mie = int(user["-ANG.Aer.NbGauss"])
    got = A.aerosols(p, user["-SOS_Main.Wa"], user["-AER.AOTref"], nb_mie, 2 * nb_mie, at_waref=False)
    _coefficients_digit_for_digit(got, g)
    assert abs(got["a_tronc"] - float(g["aer_a_tronc"])) <= 1.001e-5 and abs(got["piztr"] - float(g["aer_piztr"])) <= 1.001e-5   # F9.5
    assert np.allclose([got["kmat1"], got["kmat2"]], g["kmat"], rtol=6e-5)          # printed E13.5: five digits
    if "coef_tronca" in g.files and float(g["coef_tronca"]) != 0.0:
        assert abs(got["coef_tronca"] - float(g["coef_tronca"])) <= 1e-12


MODEL_CASES = ["wmo_continental", "wmo_user_865", "sf_maritime_rh70", "sf_urban_rh0", "ext_phase_fct", "mixture_3modes_865",
               "junge_2wl_nopolar_glitter", "bilnd_vc1_2wl_userangles", "lnd_igmax3_breon", "lnd_osnb140"]


def _resolve(user):
    return {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v) for k, v in user.items()}


def test_component_tables_of_the_wmo_and_sf_models(pkg, monkeypatch):
    """SOS_INIT_PARAMWMO / SOS_INIT_PARAMSF on the reference's tables: hand-checked rows (0.55 um lies between the 0.5 and
    0.55 rows; relative humidity 70 % is a table row, so the modal radii are that row's)."""
    monkeypatch.setenv("SOS_ABS_ROOT", GOLD)
    A = pkg.aerosols
    r, v2, mr, mi, vol = A.init_param_wmo(0.55)
    assert r == [0.5, 0.005, 0.3, 0.0118] and abs(v2[3] - 0.30103 * np.log(10.)) < 1e-15 and vol[1] == 113.98352e-6
    assert mr == [1.53, 1.53, 1.381, 1.75] and mi[:2] == [-0.008, -0.006] and mi[3] == -0.44
    r, v2, mr, mi = A.init_param_sf(0.55, 70.0)
    assert r == [0.02846, 0.4571, 0.02911, 0.4777, 0.2041] and abs(v2[0] - 0.35 * np.log(10.)) < 1e-15
    r0, _, mr0, _ = A.init_param_sf(0.55, 0.0)
    assert r0 == [0.027, 0.43, 0.025, 0.4, 0.16] and mr0[0] == 1.53
    with pytest.raises(A.AerosolError):
        A.init_param_sf(0.55, 99.5)                          # beyond the last humidity row: the reference fails on the read
    modes = A.read_mixture_file(os.path.join(GOLD, "aer_mixture.txt"))
    assert [m["igranu"] for m in modes] == [1, 2, 1] and modes[1]["v3"] == 12.0 and abs(sum(m["rate"] for m in modes) - 1) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODEL_CASES)
def test_other_aerosol_models_vs_reference(gpu_pkg, name, monkeypatch):
    """-AER.Model 1 (WMO), 2 (Shettle & Fenn), 4 (external phase functions), 5 (user mixture) against the Aerosols.txt
    and the radiances of the compiled reference (tests/golden/aer_model_*.npz, make_golden.py aer_models).  The WMO
    dust-like component runs Mie theory up to size parameter 4000 (HBM-scratch form of k_mie)."""
    import cases
    monkeypatch.setenv("SOS_ABS_ROOT", GOLD)
    rs, A = gpu_pkg.run_sos, gpu_pkg.aerosols
    g = np.load(os.path.join(GOLD, "aer_model_%s.npz" % name))
    user = _resolve(json.loads(str(g["user_json"])))
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    rs.validate_parameters(p)          # also fills the reference-wavelength indices of a single-wavelength run (SOS_PROC.F:1704)
    nb_mie = int(user["-ANG.Aer.NbGauss"])
    got = A.aerosols(p, user["-SOS_Main.Wa"], 0.1, nb_mie, 2 * nb_mie, at_waref=user["-SOS_Main.Wa"] == user["-AER.Waref"])
    _coefficients_digit_for_digit(got, g)
    assert abs(got["a_tronc"] - float(g["aer_a_tronc"])) <= 1.001e-5 and abs(got["piztr"] - float(g["aer_piztr"])) <= 1.001e-5
    assert np.allclose([got["kmat1"], got["kmat2"]], g["kmat"], rtol=6e-5)
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    # 1e-9 like every other end-to-end golden (2e-7 over land: REAL*4 surface matrices, tests/test_land.py)
    cases.compare_proc_outputs(rs, out, g, rtol=2e-7 if int(user.get("-SURF.Type", 0)) >= 3 else 1e-9)


@pytest.mark.gpu
def test_sos_proc_with_its_own_aerosol_model(gpu_pkg):
    """-AER.Model 0 end to end (BASELINE config 2 with the LND model: Mie kernel, size distribution, truncated expansion,
    profile, solve, recomposition): radiances within 1e-9 of the reference run."""
    import cases
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_cfg2_lnd_lambert.npz"))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    cases.compare_proc_outputs(rs, out, g, rtol=1e-9)


def test_vectorised_decompo_equals_the_loop_restatement(pkg):
    """decompo_legendre (sequential sums over whole arrays) == tests/aerosol_loops.py (the statement-for-statement loops), bit
    for bit, with and without truncation, spherical and non-spherical (P22 != P11) input."""
    import aerosol_loops
    A = pkg.aerosols
    rng = np.random.default_rng(1)
    for nbm, osnb, itr, g in ((24, 48, 1, 0.6), (24, 48, 0, 0.85), (5, 10, 1, 0.7), (40, 80, 1, 0.8)):
        xmu, xhr = A.mie_angles(nbm)
        l = np.arange(3 * osnb)
        p11 = np.maximum(np.polynomial.legendre.legval(xmu, (2 * l + 1) * g ** l), 1e-6)
        p12 = 0.1 * p11 * (1 - xmu ** 2) * rng.uniform(0.5, 1.5, len(xmu))
        p22, p33 = p11 * rng.uniform(0.9, 1.0, len(xmu)), 0.9 * p11 * xmu
        a = A.decompo_legendre(itr, xmu, xhr, osnb, p11, p12, p22, p33)
        b = aerosol_loops.decompo_legendre_loops(itr, xmu, xhr, osnb, p11, p12, p22, p33)
        for k in ("alpha", "beta", "gamma", "zeta", "beta22", "delta33"):
            assert np.array_equal(a[k], b[k]), (nbm, itr, k)
        assert a["coef_tronca"] == b["coef_tronca"] and a["itronc"] == b["itronc"]


def test_batched_decompo_equals_the_single_form(pkg):
    """decompo_legendre_many (the Legendre expansions of a chunk of wavelengths in one pass, sos_spectrum) == decompo_legendre
    per phase function, bit for bit: truncated, truncation refused (the retry without it), no truncation asked."""
    A = pkg.aerosols
    rng = np.random.default_rng(1)
    for nbm, osnb in ((12, 24), (16, 32), (40, 80)):
        xmu, xhr = A.mie_angles(nbm)
        w = len(xmu)
        th = np.arccos(np.clip(xmu, -1, 1))
        rows = []
        for b in range(9):
            g = rng.uniform(0.0, 0.9) if b else 0.0                  # b = 0: isotropic -> the truncation is refused
            hg = (1 - g * g) / (1 + g * g - 2 * g * xmu) ** 1.5
            p11 = hg * (1 + 0.01 * rng.random(w))
            rows.append((p11, -0.1 * hg * np.sin(th) ** 2 * rng.random(), p11 * (1 - 0.01 * rng.random(w)), hg * xmu * rng.random()))
        stack = [np.array([r[k] for r in rows]) for k in range(4)]
        for itr in (0, 1):
            many = A.decompo_legendre_many(itr, xmu, xhr, osnb, *stack)
            kept = 0
            for b, r in enumerate(rows):
                one = A.decompo_legendre(itr, xmu, xhr, osnb, *r)
                for k in one:
                    assert np.array_equal(np.asarray(one[k]), np.asarray(many[b][k])), (nbm, itr, b, k)
                kept += one["itronc"]
            assert (kept > 0) == bool(itr) and kept < len(rows)
