"""Parity of the sea-surface and azimuth-recomposition pieces.
CPU part: the host routine of the C ABI (SOS_MAT_FRESNEL) and the oracle against golden vectors.
GPU part: sosgpu_glitter / sosgpu_trphi against the oracle and the goldens."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _angles(pkg, ng=12):
    return pkg.synth.gauss_angles(ng, 35.0)


def test_mat_fresnel_host_vs_oracle(pkg, oracle):
    """sosgpu_mat_fresnel_host is host code (no GPU needed): bit-identical to the restatement."""
    for ng, os_ns, ind in [(12, 24, 1.34), (24, 48, 1.34), (40, 80, 1.33)]:
        mu, w, n0 = _angles(pkg, ng)
        got = pkg.surface.mat_fresnel(mu, w, ind, os_ns)
        ref = oracle.glitter(mu, w, 5.0, ind, 2, os_ns, os_ns + 2)["coef"]
        assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_oracle_glitter_vs_golden(pkg, oracle):
    g = np.load(os.path.join(GOLD, "glitter_n13.npz"))
    for wind in (2.0, 7.0):
        o = oracle.glitter(g["mu"], g["chr"], wind, 1.34, 24, 24, 48)
        assert np.array_equal(o["il"], g["il_w%d" % wind])
        assert np.abs(o["e"] - g["e_w%d" % wind]).max() <= 1e-14 * np.abs(g["e_w%d" % wind]).max()
        assert np.array_equal(o["rsurf"], g["rsurf_w%d" % wind])


def test_oracle_trphi_vs_golden(oracle):
    g = np.load(os.path.join(GOLD, "trphi_n13.npz"))
    for i in range(int(g["ncases"])):
        kw = dict(igli=int(g["igli%d" % i]), wind=float(g["wind%d" % i]), ifresnel=int(g["ifresnel%d" % i]),
                  ipolar=int(g["ipolar%d" % i]), n0=int(g["n0"]))
        for k, phi in enumerate(g["phis"]):
            out = oracle.trphi(g["mu"], g["rec"], 0.4, 0.05, float(phi), **kw)
            exp = g["out%d" % i][k]
            for q in range(4):
                assert np.abs(out[q] - exp[q]).max() <= 1e-13 * max(1.0, np.abs(exp[q]).max()), (i, k, q)


def test_oracle_polar_branches(oracle):
    """SOS_POLAR (SOS_TRPHI.F:1865-1903): all branches incl. the undefined value -999."""
    assert oracle.polar(1.0, 0.0, 0.0) == (-999.0, 0.0, 0.0)
    assert oracle.polar(0.0, 0.0, 0.5)[0] == 45.0 and oracle.polar(0.0, 0.0, 0.5)[1] == -999.0
    assert oracle.polar(1.0, 0.0, -0.5)[0] == -45.0
    a = oracle.polar(2.0, 0.3, 0.3)
    assert abs(a[0] - 22.5) < 1e-12 and abs(a[1] - 100 * np.hypot(0.3, 0.3) / 2) < 1e-12
    assert abs(oracle.polar(2.0, -0.3, 0.3)[0] - 67.5) < 1e-12
    assert abs(oracle.polar(2.0, -0.3, -0.3)[0] + 67.5) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("wind", [2.0, 7.0])
def test_glitter_gpu_vs_oracle_and_golden(gpu_pkg, oracle, wind):
    """Cox-Munk matrices on the GPU: series lengths IL identical, E(0:IL) to 1e-12, REAL*4 matrices equal to
    the reference's up to rare last-bit flips (device libm rounds exp/cos differently by <= 1 ulp of fp64)."""
    g = np.load(os.path.join(GOLD, "glitter_n13.npz"))
    mu, chr_ = g["mu"], g["chr"]
    out = gpu_pkg.surface.glitter_matrices(mu, chr_, wind, 1.34, 24, 24, 48)
    il = out["il"].cpu().numpy()
    e = out["e"].cpu().numpy()
    rs = out["rsurf"].cpu().numpy()
    ref = oracle.glitter(mu, chr_, wind, 1.34, 24, 24, 48)
    assert np.array_equal(il, ref["il"]) and np.array_equal(il, g["il_w%d" % wind])
    assert np.abs(e - ref["e"]).max() <= 1e-12 * np.abs(ref["e"]).max()
    for expect in (ref["rsurf"], g["rsurf_w%d" % wind]):
        diff = rs != expect
        assert diff.mean() < 1e-3, diff.mean()
        scale = np.abs(expect).max()
        assert np.abs(rs.astype(np.float64) - expect).max() <= 2e-7 * scale


@pytest.mark.gpu
def test_glitter_gpu_full_size(gpu_pkg, oracle):
    """BASELINE config 4 size: N=41, OS_NB=80, OS_NS=80, OS_NM=160, wind 7 m/s."""
    mu, w, n0 = gpu_pkg.synth.gauss_angles(40, 35.0)
    out = gpu_pkg.surface.glitter_matrices(mu, w, 7.0, 1.34, 80, 80, 160)
    ref = oracle.glitter(mu, w, 7.0, 1.34, 80, 80, 160)
    assert np.array_equal(out["il"].cpu().numpy(), ref["il"])
    rs = out["rsurf"].cpu().numpy()
    assert (rs != ref["rsurf"]).mean() < 1e-3
    assert np.abs(rs.astype(np.float64) - ref["rsurf"]).max() <= 2e-7 * np.abs(ref["rsurf"]).max()


@pytest.mark.gpu
def test_trphi_gpu_vs_oracle(gpu_pkg, oracle):
    import torch
    g = np.load(os.path.join(GOLD, "trphi_n13.npz"))
    mu, n0, rec, phis = g["mu"], int(g["n0"]), g["rec"], g["phis"]
    S = gpu_pkg.synth
    w = S.gauss_angles(12, 35.0)[1]
    al, be, ga, ze = S.hg_phase(24, 0.6)
    for i in range(int(g["ncases"])):
        igli, wind, ifres, ipol = int(g["igli%d" % i]), float(g["wind%d" % i]), int(g["ifresnel%d" % i]), int(g["ipolar%d" % i])
        cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, ifresnel=ifres, ipolar=ipol, ind_surf=1.34)
        out = cx.trphi(torch.from_numpy(rec), rec.shape[0], 0.4, 0.05, phis, igli=igli, wind=wind).cpu().numpy()
        for k, phi in enumerate(phis):
            ref = oracle.trphi(mu, rec, 0.4, 0.05, float(phi), igli=igli, n0=n0, wind=wind, ifresnel=ifres, ipolar=ipol)
            for q in range(4):
                tol = 1e-9 * np.abs(ref[q]) + 1e-12 * max(1.0, np.abs(ref[q]).max())
                assert np.all(np.abs(out[k, q] - ref[q]) <= tol), (i, k, q, np.abs(out[k, q] - ref[q]).max())
                assert np.all(np.abs(out[k, q] - g["out%d" % i][k][q]) <= tol)
            n = len(mu)
            for jj in list(range(0, n)) + list(range(n + 1, 2 * n + 1)):
                xan, tpol, lpol = oracle.polar(out[k, 0, jj], out[k, 1, jj], out[k, 2, jj])
                assert abs(out[k, 4, jj] - xan) <= 1e-9 * max(1.0, abs(xan))
                assert abs(out[k, 5, jj] - tpol) <= 1e-9 * max(1.0, abs(tpol))
                assert abs(out[k, 6, jj] - lpol) <= 1e-12 + 1e-9 * abs(lpol)
        cx.close()
