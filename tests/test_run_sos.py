"""The drop-in boundary: radiativetransfer-sos_amd/run_sos.py mirrors binding/run_sos.py.
CPU: parameter surface (defaults, merge, positional order, keyword names) and the host-side restatements of the
steps upstream of the hot path (angles, Rayleigh optical thickness, no-gas profile).
GPU: sos_proc(**kwargs) end to end against the reference SOS_PROC outputs (tests/golden/sos_proc_*.npz)."""
import json
import os

import numpy as np
import pytest

import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROC_CASES = ["cfg1_lambert", "glitter_polar", "flatsea_zout", "nopolar_polar"]


def test_parameter_surface(pkg):
    rs = pkg.run_sos
    d = rs.default_parameters()
    # every key of the reference dictionary (binding/run_sos.py:459-559), incl. the three never forwarded
    assert len(d) == 97 and d["-SOS.IGmax"] == 200 and d["-SOS.View"] == 2 and d["-AER.Tronca"] == 1
    assert d["-SOS_Main.Wa"] == -999.0 and d["-ANG.Rad.NbGauss"] == -999 and d["-AER.MMD.JD.rmax"] == 50.0
    assert d["-SURF.File"] == "DEFAULT" and d["-SOS.MDF"] == 0.0279
    u = rs.update_parameters(d, {"-SOS_Main.Wa": 0.44, " -SURF.Alb": 0.02, "-NOT.A.KEY": 1})
    assert u["-SOS_Main.Wa"] == 0.44 and " -SURF.Alb" not in u and "-NOT.A.KEY" not in u   # unknown keys dropped
    t = rs.set_sos_params(u, trace=True)
    assert len(t) == 96 and t[-1] is True and t[-2] == 0 and t[2] == 0.44
    kw = rs.sos_proc_kwargs(u)
    assert list(kw)[:6] == ["resroot", "ficmain_log", "wa_simu", "nbmu_gauss_lum", "ficangles_user_lum", "tetas"]
    assert list(kw)[-8:] == ["igmax", "ipolar", "itrphi", "phios", "pas_phi", "imode_ckd_calcul", "ier", "trace"]
    assert len(rs.OUTPUT_NAMES) == 23


def test_host_restatements_against_reference_outputs(pkg):
    """angles / Rayleigh tau / profile restatements reproduce what the reference SOS_PROC reported for cfg 1:
    25 directions, theta table, Rayleigh optical thickness 9.480316896194211E-02 (reference stdout)."""
    rs = pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_cfg1_lambert.npz"))
    mu, ga, n0, ind = rs.angles(24, 35.0)
    assert len(mu) == int(g["nblum"]) == 25 and abs(mu[n0 - 1] - np.cos(np.radians(35.0))) < 1e-14
    assert np.allclose(np.degrees(np.arccos(mu)), g["vza"][:25], rtol=0, atol=1e-11)
    assert abs(np.sum(ga) - 1.0) < 1e-13 and ga[n0 - 1] == 0.0
    tr = rs.rayleigh_optical_thickness(0.550, 1013.0)
    assert abs(tr - 9.480316896194211e-02) < 1e-16
    h, xdel, ydel, z = rs.profile_nogas(tr, 8.0, 0.0, 2.0)
    assert len(h) == 102 and h[0] == 0.0 and abs(h[-1] - tr) < 1e-8 * tr and np.all(xdel == 0) and np.all(ydel == 1)
    assert z[0] == 120.0 and np.all(np.diff(z) < 0)
    # with aerosols: levels by bisection, fractions sum to 1, monotone optical depth
    h, xdel, ydel, z = rs.profile_nogas(0.0948, 8.0, 0.3, 2.0)
    assert abs(h[-1] - 0.3948) < 1e-8 and np.all(np.diff(h) > 0) and np.allclose(xdel[1:] + ydel[1:], 1.0, atol=1e-7)


def test_sos_proc_rejects_bad_or_unsupported_parameters(pkg):
    rs = pkg.run_sos
    base = {"-SOS_Main.Wa": 0.55, "-ANG.Thetas": 30.0, "-AP.AbsProfile.Type": 7, "-AER.AOTref": 0.0, "-SURF.Alb": 0.1,
            "-AP.HR": 8.0, "-SOS.View": 1, "-SOS.View.Phi": 0.0}
    kw = lambda **over: rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), dict(base, **over)))
    with pytest.raises(rs.SosProcError):                       # gas absorption needs the spectral resolution of the CKD tables
        rs.sos_proc(**kw(**{"-AP.AbsProfile.Type": 2, "-AP.Psurf": 1013.0}))
    with pytest.raises(rs.SosProcError):
        rs.sos_proc(**kw(**{"-AP.AbsProfile.Type": 9}))
    with pytest.raises(rs.SosProcError):                       # aerosols asked for, -AER.Model left undefined
        rs.sos_proc(**kw(**{"-AER.AOTref": 0.3}))
    with pytest.raises(rs.SosProcError):                       # aerosol layer profile without its two altitudes
        rs.sos_proc(**kw(**{"-AP.AerProfile.Type": 2}))
    with pytest.raises(rs.SosProcError):                       # ... and it excludes gas absorption (SOS_PROC.F:2352)
        rs.sos_proc(**kw(**{"-AP.AerProfile.Type": 2, "-AP.AerLayer.Zmin": 1.0, "-AP.AerLayer.Zmax": 2.0,
                            "-AP.AbsProfile.Type": 2, "-AP.Psurf": 1013.0, "-AP.SpectralResol": 10}))
    with pytest.raises(TypeError):
        rs.sos_proc(wa_simu=0.55)


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROC_CASES)
def test_sos_proc_vs_reference(gpu_pkg, name):
    """run_sos.sos_proc(**kwargs) == reference sos.sos_proc outputs: I,Q,U tables to 1e-9 relative (plus 1e-12 of
    the I scale for near-zero Q/U), angles/flux scalars to 1e-9; identical table shapes and fill pattern."""
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    kw = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    out = rs.sos_proc(**kw)
    cases.compare_proc_outputs(rs, out, g)


def test_parameter_validation_matches_the_reference_error_numbers(pkg):
    """run_sos.validate_parameters raises, for each of the 69 broken keyword sets of tests/golden/validation.json, the error
    number the compiled reference's SOS_PROC printed for the same set (make_golden.py validation: one reference run per set,
    `ERROR_<n>` parsed from its standard output).  Code 0 = the reference accepted the set; 4700 = it failed later, inside
    SOS: validate_parameters lets those through; -6 = its 'Nadal's BPDF model is not supported' exit, taken at the same place."""
    rs = pkg.run_sos
    cases_ = json.load(open(os.path.join(GOLD, "validation.json")))
    assert len(cases_) >= 69
    seen = set()
    for c in cases_:
        user = {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v) for k, v in c["user"].items()}
        user["-SOS_Main.ResRoot"] = "/tmp/unused"
        p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
        if c["code"] in (0, 4700):
            rs.validate_parameters(dict(p))
            continue
        if c["code"] == -6:
            with pytest.raises(rs.SosProcError, match="Nadal"):
                rs.validate_parameters(dict(p))
            continue
        with pytest.raises(rs.SosProcError) as e:
            rs.validate_parameters(dict(p))
        assert e.value.code == c["code"], (c["code"], e.value.code, c["user"])
        assert "ERROR_%d" % c["code"] in str(e.value)
        seen.add(c["code"])
    assert len(seen) >= 60
    # single-wavelength side effect (SOS_PROC.F:1704-1707): the reference-wavelength indices default to the simulation ones
    user = dict(cases_[17]["user"], **{"-SOS_Main.Wa": 0.55})
    p = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    rs.validate_parameters(p)
    assert p["rn_waref"] == p["rn_wa"] == 1.45 and p["in_waref"] == p["in_wa"]


def test_angle_and_aerosol_text_files_match_the_reference(pkg, tmp_path):
    """SOS_UsedAngles.txt, Aer_UsedAngles.txt (user angles in both sets) and the Aerosols.txt of an aerosol-free run, byte for
    byte as the compiled reference wrote them (tests/golden/angle_files.json, make_golden.py angle_files); only the padded
    user-file name differs."""
    rs = pkg.run_sos
    d = json.load(open(os.path.join(GOLD, "angle_files.json")))
    uf = str(tmp_path / "user_ang.txt")
    open(uf, "w").write("".join("%.1f\n" % a for a in d["user_angles_deg"]))
    nb_lum, nb_mie = d["user"]["-ANG.Rad.NbGauss"], d["user"]["-ANG.Aer.NbGauss"]
    os_nb, os_ns = 2 * nb_mie, 2 * nb_lum
    mu, ga, n0, ind = rs.angles(nb_lum, d["user"]["-ANG.Thetas"], uf)
    rs.write_used_angles(str(tmp_path / "lum.txt"), mu, ga, n0, ind, nb_lum, d["user"]["-ANG.Thetas"], os_nb, os_ns, os_nb + os_ns, uf)
    rs.write_mie_angles(str(tmp_path / "mie.txt"), nb_mie, os_nb, uf)
    z = np.zeros(os_nb + 1)
    rs.write_aerosols_file(str(tmp_path / "aer.txt"), dict(alpha=z, beta=z, gamma=z, zeta=z, a_tronc=0.0, piztr=0.0, piz=0.0))

    def lines(text):
        return [ln.rstrip() for ln in text.replace(uf, "@USERANG").splitlines()]
    assert lines(open(tmp_path / "lum.txt").read()) == lines(d["files"]["SOS_UsedAngles.txt"])
    assert lines(open(tmp_path / "mie.txt").read()) == lines(d["files"]["Aer_UsedAngles.txt"])
    assert lines(open(tmp_path / "aer.txt").read()) == lines(d["files"]["Aerosols.txt"])


def test_ascii_result_files_layout(pkg, tmp_path):
    """SOS_Up.txt / SOS_Down.txt in the layout of the reference's PYTHON front end (gen_sos_output, binding/run_sos.py:196-317;
    the script itself cannot run here -- it imports its f2py extension -- so its numbers are pinned against the files of the
    Fortran main in test_main_ascii_writer_reproduces_the_reference_files_byte_for_byte).  Layout: fixed-azimuth view = the
    phi + 180 half plane first (negated zenith angles, descending), then the phi half plane; polar view = one block of NBLUM
    rows per azimuth, ceil(360 / dphi) blocks; numbers in the %13.6e / %7.2f columns of the reference's format strings."""
    rs = pkg.run_sos
    nbl = 5
    rng = np.random.default_rng(3)
    t = lambda: rng.uniform(0.01, 1.0, (361, 81))
    vza = np.zeros(81); vza[:nbl] = [5., 20., 40., 60., 80.]
    phi = np.zeros(361); phi[:5] = [0., 80., 160., 240., 320.]
    arrs = [t() for _ in range(7)]
    rs.gen_sos_output(str(tmp_path), 1, 1, -1.0, nbl, -999, phi, vza, *arrs)
    lines = open(tmp_path / "SOS_Up.txt").read().splitlines()
    data = [ln for ln in lines if not ln.startswith("#")]
    assert len(data) == 2 * nbl and lines[0].startswith("#UPWARD RADIANCE FIELD")
    z = [float(ln.split()[0]) for ln in data]
    assert z == [-80., -60., -40., -20., -5., 5., 20., 40., 60., 80.]
    assert abs(float(data[0].split()[2]) - arrs[1][0, nbl - 1]) <= 5e-7 * arrs[1][0, nbl - 1]       # I of the phi + 180 plane
    assert abs(float(data[nbl].split()[2]) - arrs[1][1, 0]) <= 5e-7 * arrs[1][1, 0]                 # I of the phi plane
    assert len(data[0]) == len("  %7.2f %7.2f  %13.6e  %13.6e  %13.6e  %7.2f %7.2f %13.6e" % ((1.0,) * 8))
    rs.gen_sos_output(str(tmp_path), 2, 2, 2.5, nbl, 80, phi, vza, *arrs)
    lines = open(tmp_path / "SOS_Down.txt").read().splitlines()
    data = [ln for ln in lines if not ln.startswith("#")]
    assert len(data) == 5 * nbl and lines[0].startswith("#DOWNWARD RADIANCE FIELD")
    assert [float(ln.split()[0]) for ln in data[::nbl]] == [0., 80., 160., 240., 320.]
    assert any("2.5" in ln and "altitude" in ln for ln in lines)


MAIN_ASCII = ["plane", "polar"]


def _main_ascii_user(g):
    return {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v)
            for k, v in json.loads(str(g["user_json"])).items()}


def _data_rows(raw):
    return np.array([[float(x) for x in ln.split()] for ln in raw.decode("latin-1").splitlines() if not ln.startswith("#")])


@pytest.mark.parametrize("name", MAIN_ASCII)
def test_main_ascii_writer_reproduces_the_reference_files_byte_for_byte(pkg, tmp_path, name):
    """SOS_Up.txt / SOS_Down.txt (+ the .UserAng pair) as the reference's command-line program writes them
    (SOS_ABS_MAIN.F:2250-2444, formats 55/56, headers SOS_TRPHI.F:1570-1796): tests/golden/main_ascii_*.npz hold the bytes of
    the files SOS_ABS_MAIN.exe wrote (make_golden.py main_ascii) and the 23 outputs of the reference's sos_proc_ for the same
    keywords.  Fed those outputs, run_sos.write_main_output reproduces the files byte for byte -- headers (with the truncated
    degree-sign lines), F7.2 / E13.6 columns, plane ordering, the row-IPHI quirk of the polar upward file."""
    rs = pkg.run_sos
    g = np.load(os.path.join(GOLD, "main_ascii_%s.npz" % name))
    user = _main_ascii_user(g)
    out = tuple(g[nm] if g[nm].shape else g[nm][()] for nm in rs.OUTPUT_NAMES)
    out = (int(out[0]),) + out[1:]
    pu, pd, uu, ud = (str(tmp_path / f) for f in ("up", "down", "up_user", "down_user"))
    has_user = "file_up_user" in g.files
    rs.write_main_output(pu, pd, out, int(user["-SOS.View"]), user.get("-SOS.View.Phi", -999.0), user.get("-SOS.View.Dphi", -999),
                         float(user.get("-SOS.OutputAlt", -1.0)), uu if has_user else None, ud if has_user else None)
    for key, path in (("up", pu), ("down", pd)) + ((("up_user", uu), ("down_user", ud)) if has_user else ()):
        ref = g["file_" + key].tobytes()
        got = open(path, "rb").read()
        assert got == ref, (key, [(a, b) for a, b in zip(got.splitlines(), ref.splitlines()) if a != b][:3])
    assert len(_data_rows(g["file_up"].tobytes())) == (2 * out[0] if name == "plane" else 9 * out[0])
    # the reference's Python front end formats the same numbers with C-style %13.6e: gen_sos_output's numbers equal the
    # Fortran file's to the digits both carry
    rs.gen_sos_output(str(tmp_path), int(user["-SOS.View"]), 1, 0.0, out[0], user.get("-SOS.View.Dphi", -999), out[2], out[3],
                      out[4], out[5], out[6], out[7], out[8], out[9], out[10])
    mine = _data_rows(open(tmp_path / "SOS_Up.txt", "rb").read())
    ref = _data_rows(g["file_up"].tobytes())
    n = min(len(mine), len(ref))                            # (polar view: the script writes ceil(360/dphi) blocks, the main 360/dphi + 1)
    assert n >= 8 * out[0] or name == "plane"
    sca = 2 if name == "polar" else 1
    cols = [c for c in range(ref.shape[1]) if not (name == "polar" and c == sca)]      # (row-IPHI quirk of the main's SCA_ANG)
    assert np.allclose(mine[:n][:, cols], ref[:n][:, cols], rtol=2e-6, atol=0.0051)


@pytest.mark.gpu
@pytest.mark.parametrize("name", MAIN_ASCII)
def test_main_ascii_files_from_the_gpu_run(gpu_pkg, tmp_path, name):
    """The same files from run_sos.sos_proc on the GPU: headers identical, every data column equal to the reference file's to
    one unit of its last printed digit (the values agree to 1e-9; a digit flips where the 7th decimal sits on a rounding
    boundary), and at least 98 % of the lines byte-identical."""
    rs = gpu_pkg.run_sos
    g = np.load(os.path.join(GOLD, "main_ascii_%s.npz" % name))
    user = _main_ascii_user(g)
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    cases.compare_proc_outputs(rs, out, g)
    pu, pd = str(tmp_path / "up"), str(tmp_path / "down")
    rs.write_main_output(pu, pd, out, int(user["-SOS.View"]), user.get("-SOS.View.Phi", -999.0), user.get("-SOS.View.Dphi", -999),
                         float(user.get("-SOS.OutputAlt", -1.0)))
    for key, path in (("up", pu), ("down", pd)):
        ref, got = g["file_" + key].tobytes().splitlines(), open(path, "rb").read().splitlines()
        assert len(ref) == len(got)
        assert [ln for ln in ref if ln.startswith(b"#")] == [ln for ln in got if ln.startswith(b"#")]
        same = sum(a == b for a, b in zip(ref, got))
        assert same >= 0.98 * len(ref), (key, same, len(ref))
        a, b = _data_rows(b"\n".join(got)), _data_rows(b"\n".join(ref))
        ncol = a.shape[1]
        ang = [c for c in range(ncol) if c not in (ncol - 6, ncol - 5, ncol - 4, ncol - 1)]
        assert np.all(np.abs(a[:, ang] - b[:, ang]) <= 0.0101)                                  # F7.2 columns
        ecol = [ncol - 6, ncol - 5, ncol - 4, ncol - 1]
        assert np.all(np.abs(a[:, ecol] - b[:, ecol]) <= 1.01e-6 * np.abs(b[:, ecol]) + 1e-30)  # E13.6 columns


RANDOM_CASES = sorted(f[len("sos_proc_"):-4] for f in os.listdir(GOLD) if f.startswith("sos_proc_rand_"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", RANDOM_CASES)
def test_sos_proc_random_keyword_sets_vs_reference(gpu_pkg, name, tmp_path, monkeypatch):
    """Fuzz at the drop-in boundary: seeded random keyword sets (surfaces 0-5 and 7, exponential and layer aerosol profiles,
    scalar and polarised runs, IGMAX limits, output altitudes, user viewing angles, fixed-azimuth and polar views; rand_12..19, 32..39:
    multi-bin CKD bands with random atmospheres, gas amounts and both CKD modes) through the
    compiled reference's SOS_PROC (make_golden.py proc_random) and through run_sos.sos_proc on the GPU.  Aerosols enter through
    the reference's own Aerosols.txt (-AER.UserFile), so the comparison is at 1e-9 (2e-7 over land: REAL*4 surface matrices)."""
    rs = gpu_pkg.run_sos
    monkeypatch.setenv("SOS_ABS_ROOT", GOLD)               # trimmed CKD tables of the two fixture wavenumbers
    g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
    user = {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v)
            for k, v in json.loads(str(g["user_json"])).items()}
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT"})
    coef = None
    if user["-AER.AOTref"] != 0.0:
        f = str(tmp_path / "Aerosols_user.txt")
        rs.write_aerosols_file(f, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")}, *g["kmat"])
        user["-AER.UserFile"] = f
        coef = 0.0
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=2e-7 if int(user["-SURF.Type"]) >= 3 else 1e-9)


@pytest.mark.gpu
def test_sos_proc_many_threads_and_streams_vs_reference(gpu_pkg, tmp_path, monkeypatch):
    """run_sos.sos_proc_many: the 20 random keyword sets (each repeated twice: 40 calls) issued from 6 host threads on their own
    HIP streams, every one writing its result files into its own directory -- each call's 23 outputs still match the compiled
    reference's, and the two copies of a call are bit-identical (no cross-talk between concurrent wavelengths)."""
    rs = gpu_pkg.run_sos
    monkeypatch.setenv("SOS_ABS_ROOT", GOLD)
    kws, golds, coefs = [], [], []
    for rep in range(2):
        for name in RANDOM_CASES:
            g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
            user = {k: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v)
                    for k, v in json.loads(str(g["user_json"])).items()}
            d = tmp_path / ("%s_%d" % (name, rep))
            d.mkdir()
            user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT", "-SOS_Main.ResRoot": str(d)})
            coef = None
            if user["-AER.AOTref"] != 0.0:
                f = str(d / "Aerosols_user.txt")
                rs.write_aerosols_file(f, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")},
                                       *g["kmat"])
                user["-AER.UserFile"] = f
                coef = 0.0
            kws.append(rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
            golds.append((g, int(user["-SURF.Type"])))
            coefs.append(coef)
    outs = rs.sos_proc_many(kws, n_workers=6)
    assert len(outs) == len(kws)
    for out, (g, isurf), coef in zip(outs, golds, coefs):
        cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=2e-7 if isurf >= 3 else 1e-9)
    n = len(RANDOM_CASES)
    for i in range(n):
        for a, b in zip(outs[i], outs[i + n]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), RANDOM_CASES[i]
    with pytest.raises(rs.SosProcError):                   # a failing call surfaces after the others have ended
        bad = dict(kws[0]); bad["isurf"] = 6
        rs.sos_proc_many([kws[1], bad, kws[2]], n_workers=2)
