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


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_sos_os_vs_golden(gpu_pkg, name):
    """HIP path against the committed outputs of the reference Fortran itself."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sos_os_%s.npz" % name))
    case = cases.make_case(name)
    got = cases.run_gpu(gpu_pkg, case)
    for b, r in enumerate(got):
        assert np.array_equal(r["ig_counts"], g["ig%d" % b]), (name, b)
        cases.compare_records(r["records"], g["rec%d" % b], 1e-9, "%s bin %d" % (name, b))
        assert np.allclose([r["emoins"], r["eplus"]], g["flux%d" % b], rtol=1e-9, atol=0)


def _batch(gpu_pkg, nb, nt=30, seed=5):
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(40, 35.0)
    al, be, ga, ze = S.hg_phase(80, 0.75)
    bins = S.ckd_bins(nb, nt, seed=seed)
    h, x, y, iborm = S.rescale_profile(bins["h"], bins["xdel"], bins["ydel"], 0.0, 0.95, 0.95, 80)
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.1)
    return cx, (mu, w, n0, al, be, ga, ze), (h, x, y, bins["zprof"]), bins["aik"], iborm


def test_aggregate_vs_oracle(gpu_pkg, oracle):
    """sosgpu_aggregate == serial SOS_AGGREGATE over per-bin oracle results (8 bins, N=41)."""
    import torch
    nb = 8
    cx, (mu, w, n0, al, be, ga, ze), (h, x, y, z), aik, iborm = _batch(gpu_pkg, nb)
    out = cx.solve(cx.upload_bins(h, x, y))
    scal = np.zeros((nb, 4))
    scal[:, 1] = h[:, -1]
    scal[:, 2] = h[:, -1]
    rec, sc = cx.aggregate(out, aik, scal=scal)
    torch.cuda.synchronize()
    recs, nf, sb = np.zeros((nb, iborm + 1, 3, 2 * len(mu) + 1)), np.zeros(nb, dtype=np.int32), np.zeros((nb, 7))
    for b in range(nb):
        r = oracle.sos_os(mu, w, 80, h[b], x[b], y[b], al, be, ga, ze, n0=n0, ro=0.1, iborm=iborm)
        nf[b] = len(r["records"])
        recs[b, :nf[b]] = r["records"]
        sb[b] = [0.0, r["emoins"], r["eplus"], h[b, -1], h[b, -1], 0.0, 0.0]
    exp_rec, exp_scal = oracle.aggregate(recs, nf, aik, sb)
    got = rec[0].cpu().numpy()
    cases.compare_records(got[:len(exp_rec)], exp_rec, 1e-9, "aggregate")
    assert np.all(got[len(exp_rec):] == 0)
    fin = gpu_pkg.dist.finish_scalars(sc)
    assert np.allclose([fin["emoins"][0], fin["eplus"][0]], exp_scal[1:3], rtol=1e-9)
    assert np.allclose([fin["ttot_tronc"][0], fin["ttot_vrai"][0]], exp_scal[3:5], rtol=1e-9)
    assert fin["n_orders"][0] == nf.max()
    cx.close()


def test_full_size_properties(gpu_pkg):
    """BASELINE-size batch (1024 bins, N=41, NT=30, OS_NB=80): size-independent properties.
    (1) determinism: two launches are bit-identical; (2) a permuted batch gives the permuted result
    bit for bit (bins are independent); (3) aggregation is linear: aggregate(aik) == sum aik*rec to
    fp64 noise and sharded partial sums add up to the unsharded aggregate."""
    import torch
    nb = 1024
    cx, _, (h, x, y, z), aik, iborm = _batch(gpu_pkg, nb, seed=9)
    bins = cx.upload_bins(h, x, y)
    o1 = cx.solve(bins)
    r1, n1, g1 = o1["rec"].clone(), o1["norders"].clone(), o1["iglast"].clone()
    o2 = cx.solve(bins)
    torch.cuda.synchronize()
    assert torch.equal(r1, o2["rec"]) and torch.equal(n1, o2["norders"]) and torch.equal(g1, o2["iglast"])
    assert int(n1.min()) >= 3 and int(n1.max()) <= iborm + 1 and torch.isfinite(r1).all()
    perm = np.random.default_rng(0).permutation(nb)
    o3 = cx.solve(cx.upload_bins(h[perm], x[perm], y[perm]))
    torch.cuda.synchronize()
    assert torch.equal(o3["rec"], r1[torch.from_numpy(perm).to(r1.device)])
    bs = cx.upload_bins(h, x, y, order="cost")                  # cost-sorted upload returns its permutation
    o4 = cx.solve(bs)
    torch.cuda.synchronize()
    assert torch.equal(o4["rec"], r1[torch.from_numpy(bs["perm"]).to(r1.device)])
    # linearity / sharding of the aggregate
    rec_all, sc_all = cx.aggregate(o2, aik)
    direct = (torch.from_numpy(aik).to(r1.device)[:, None, None, None] * r1).sum(0)
    scale = float(direct[:, 0].abs().max())
    assert float((rec_all[0] - direct).abs().max()) <= 1e-12 * scale
    seg = np.array([0, 300, 301, nb], dtype=np.int32)
    rec_seg, sc_seg = cx.aggregate(o2, aik, seg=seg)
    assert float((rec_seg.sum(0) - rec_all[0]).abs().max()) <= 1e-12 * scale
    assert abs(float(sc_seg[:, 6].sum()) - 1.0) < 1e-12
    # upward radiance at TOA is non-negative for the azimuthal mean (order 0, I component)
    n = cx.n
    assert float(r1[:, 0, 0, n + 1:].min()) > 0 and float(r1[:, 0, 0, :n].min()) > 0
    cx.close()


def test_malformed_bin_is_flagged_not_faulted(gpu_pkg):
    """NT outside the compiled variant is caught by the in-kernel shape guard (norders = -1)."""
    import torch
    cx, _, (h, x, y, z), aik, iborm = _batch(gpu_pkg, 2)
    bins = cx.upload_bins(h, x, y)
    bins["nt"][1] = 31   # > lp-1 = 31? lp = 32 -> nt must be <= 31 and < COLS=32; 31 is legal
    bins["nt"][1] = 40   # illegal: beyond the padded level axis
    out = cx.solve(bins)
    torch.cuda.synchronize()
    assert int(out["norders"][1]) == -1 and int(out["norders"][0]) > 0
    cx.close()


def test_failed_bin_raises_in_solve_band(gpu_pkg):
    """A malformed bin (the reference's IER = -1) must not be summed silently: solve_band raises, the aggregate skips
    the bin and flags it, and its (never written) records are not read."""
    import torch
    cx, _, (h, x, y, z), aik, iborm = _batch(gpu_pkg, 4)
    bins = cx.upload_bins(h, x, y)
    good_rec, good_fin = cx.solve_band(bins, aik)
    bins["nt"][2] = 40
    out = cx.alloc_outputs(4)
    out["rec"].fill_(float("nan"))                      # anything left unwritten must not be read
    cx.solve(bins, out)
    rec, sc = cx.aggregate(out, aik)
    fin = gpu_pkg.dist.finish_scalars(sc)
    torch.cuda.synchronize()
    assert fin["min_orders"][0] == -1 and torch.isfinite(rec).all()
    assert abs(fin["sum_aik"][0] - (aik.sum() - aik[2])) < 1e-15
    with pytest.raises(gpu_pkg.SosBinError):
        cx.solve_band(bins, aik)
    cx.close()


def test_empty_shard_and_tdifmug_aggregate(gpu_pkg):
    """A rank without bins contributes the neutral element; TDIFMUG(1..N) is aggregated next to TDIFMUS
    (SOS_AGGREGATE.F:452-459)."""
    import torch
    cx, _, (h, x, y, z), aik, iborm = _batch(gpu_pkg, 3)
    rec0, sc0 = cx.aggregate(None, None)
    torch.cuda.synchronize()
    assert float(rec0.abs().max()) == 0.0 and float(sc0[0, :7].abs().max()) == 0.0 and float(sc0[0, 8]) < -2e9
    out = cx.solve(cx.upload_bins(h, x, y))
    tdg = np.random.default_rng(2).uniform(size=(3, cx.n))
    rec, sc = cx.aggregate(out, aik, tdifmug=tdg)
    fin = gpu_pkg.dist.finish_scalars(sc)
    exp = np.zeros(cx.n)
    for b in range(3):
        exp = exp + aik[b] * tdg[b]
    assert np.array_equal(fin["tdifmug"][0], exp)
    cx.close()


def test_c_abi_reduce_single_rank(gpu_pkg):
    """sosgpu_comm_* / sosgpu_pack / sosgpu_reduce / sosgpu_unpack on a one-rank RCCL communicator: the all-reduce over
    one rank is the identity, MAX-combined elements included."""
    import ctypes as C
    import torch
    L = gpu_pkg.capi.lib()
    cx, _, (h, x, y, z), aik, iborm = _batch(gpu_pkg, 3)
    out = cx.solve(cx.upload_bins(h, x, y))
    rec, sc = cx.aggregate(out, aik)
    uid = C.create_string_buffer(128)
    gpu_pkg.capi.check(L.sosgpu_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    gpu_pkg.capi.check(L.sosgpu_comm_init_rank(C.byref(comm), 1, uid, 0), "comm_init_rank")
    buf = torch.empty((1, rec[0].numel() + sc.shape[1]), dtype=torch.float64, device=rec.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    gpu_pkg.capi.check(L.sosgpu_pack(cx._h, 1, p(rec), p(sc), p(buf), st), "pack")
    gpu_pkg.capi.check(L.sosgpu_reduce(cx._h, comm, 1, p(buf), st), "reduce")
    rec2, sc2 = torch.empty_like(rec), torch.empty_like(sc)
    gpu_pkg.capi.check(L.sosgpu_unpack(cx._h, 1, p(buf), p(rec2), p(sc2), st), "unpack")
    torch.cuda.synchronize()
    assert torch.equal(rec, rec2) and torch.equal(sc, sc2)
    gpu_pkg.capi.check(L.sosgpu_comm_destroy(comm), "comm_destroy")
    cx.close()


def test_diffuse_transmissions_vs_oracle(gpu_pkg, oracle):
    """-SOS.Trans (SOS.F:600-635): order-0, black-ground solves with every direction as incidence."""
    import torch
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(8, 35.0)
    os_nb = 16
    al, be, ga, ze = S.hg_phase(os_nb, 0.6)
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=os_nb, ro=0.2)
    hs, xs, ys = [], [], []
    for k in (0.0, 0.8):
        h, x, y, z = S.profile(20, k_abs=k)
        h, x, y, ib = S.rescale_profile(h, x, y, 0.3, 0.95, 0.93, os_nb)
        hs.append(h); xs.append(x); ys.append(y)
    bins = cx.upload_bins(np.array(hs), np.array(xs), np.array(ys))
    tdifmus, tdifmug = cx.diffuse_transmissions(bins)
    torch.cuda.synchronize()
    tdifmus, tdifmug = tdifmus.cpu().numpy(), tdifmug.cpu().numpy()
    for b in range(2):
        for j in range(1, len(mu) + 1):
            r = oracle.sos_os(mu, w, os_nb, hs[b], xs[b], ys[b], al, be, ga, ze, n0=j, ro=0.0, iborm=0)
            assert abs(tdifmug[b, j - 1] - r["emoins"]) <= 1e-9 * abs(r["emoins"]) + 1e-300, (b, j)
        assert tdifmus[b] == tdifmug[b, n0 - 1]
    cx.close()


@pytest.mark.gpu
def test_solve_many_overlapping_wavelengths(gpu_pkg):
    """solver.solve_many: several wavelengths (contexts) with a few bins each on separate HIP streams give exactly the bands of
    the one-at-a-time calls (same kernels, same data: bit-identical)."""
    import torch
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(24, 35.0)
    items, ref = [], []
    for i, g in enumerate((0.6, 0.7, 0.8, 0.75, 0.65)):
        al, be, ga, ze = S.hg_phase(48, g)
        b = S.ckd_bins(6, 30, seed=50 + i)
        h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, 48)
        cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.05 * (i + 1))
        bins = cx.upload_bins(h, x, y)
        aik = torch.from_numpy(b["aik"]).to(cx.device)
        items.append((cx, bins, aik))
        rec, scal = cx.aggregate(cx.solve(bins), aik)
        torch.cuda.synchronize()
        ref.append((rec.clone(), scal.clone()))
    got = gpu_pkg.solver.solve_many(items, n_streams=3)
    torch.cuda.synchronize()
    for (r, s), (r0, s0) in zip(got, ref):
        assert torch.equal(r, r0) and torch.equal(s, s0)
    for cx, _, _ in items:
        cx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,nt,surf,zout", [(24, 30, False, False), (41, 45, False, True), (24, 110, False, False),
                                            (13, 30, True, False), (13, 97, True, True), (49, 28, False, False)])
def test_solve_spectrum_one_launch_for_many_wavelengths(gpu_pkg, n, nt, surf, zout):
    """solver.solve_spectrum (sosgpu_os_solve_multi): the bins of several wavelengths -- different phase functions, surface
    albedos / matrices, solar directions, Fresnel flags and bin counts -- solved by ONE launch with a per-bin context give
    bit-identical records, order counts, fluxes and bands to the per-wavelength calls (LDS-resident and streamed variants)."""
    import torch
    import cases
    S = gpu_pkg.synth
    os_nb = 32
    mu, w, n0 = S.gauss_angles(n - 1, 35.0)
    assert len(mu) == n
    ctxs, bl, aiks, ref = [], [], [], []
    for i, g in enumerate((0.6, 0.8, 0.7, 0.75)):
        al, be, ga, ze = S.hg_phase(os_nb, g)
        nbw = 3 + 2 * i
        b = S.ckd_bins(nbw, nt, seed=70 + i, tau_a=0.1 + 0.1 * i)
        h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.9 + 0.02 * i, os_nb)
        kw = dict(ro=0.05 * (i + 1), ifresnel=0 if surf else i % 2)
        if surf:
            kw.update(imat_surf=1, rsurf=(1.0 + 0.3 * i) * cases._surf_matrices(n, os_nb, 7 + i))
        cx = gpu_pkg.SosContext(mu, w, 1 + (n0 - 1 + i) % n, al, be, ga, ze, iborm_max=os_nb, **kw)
        bins = cx.upload_bins(h, x, y, iborm=np.full(nbw, os_nb if i % 2 else 5, dtype=np.int32),
                              zout=1.5 if zout else -1.0, zprof=b["zprof"])
        aik = torch.from_numpy(b["aik"]).to(cx.device)
        out = cx.solve(bins)
        rec, scal = cx.aggregate(out, aik)
        torch.cuda.synchronize()
        ref.append((out, rec.clone(), scal.clone()))
        ctxs.append(cx); bl.append(bins); aiks.append(aik)
    table = gpu_pkg.solver.ContextTable(ctxs)
    bins, cob, seg = gpu_pkg.solver.concat_bins(bl)
    out = ctxs[0].alloc_outputs(bins["nb"])
    rec, scal = gpu_pkg.solver.solve_spectrum(table, bins, cob, seg, torch.cat(aiks), out=out)
    torch.cuda.synchronize()
    b0 = 0
    for i, (o, r0, s0) in enumerate(ref):
        nbw = o["rec"].shape[0]
        for k in ("norders", "iglast", "flux"):
            assert torch.equal(out[k][b0:b0 + nbw], o[k]), (i, k)
        for j in range(nbw):                               # (orders past a bin's stop hold nothing the results depend on)
            f = int(o["norders"][j])
            assert torch.equal(out["rec"][b0 + j, :f], o["rec"][j, :f]), (i, j)
        assert torch.equal(rec[i], r0[0]) and torch.equal(scal[i], s0[0]), i
        b0 += nbw
    assert int(out["norders"].min()) > 0
    for cx in ctxs:
        cx.close()


@pytest.mark.gpu
def test_context_table_rejects_mismatched_contexts(gpu_pkg):
    S = gpu_pkg.synth
    al, be, ga, ze = S.hg_phase(16, 0.6)
    mu, w, n0 = S.gauss_angles(8, 35.0)
    mu2, w2, n02 = S.gauss_angles(12, 35.0)
    a = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=16, ro=0.1)
    b = gpu_pkg.SosContext(mu2, w2, n02, al, be, ga, ze, iborm_max=16, ro=0.1)
    c = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=8, ro=0.1)
    for pair in ((a, b), (a, c)):
        with pytest.raises(gpu_pkg.capi.SosgpuError):
            gpu_pkg.solver.ContextTable(pair)
    gpu_pkg.solver.ContextTable((a, a))
    for cx in (a, b, c):
        cx.close()


STREAM_CASES = ["rayleigh_n25_nt101", "aer_n41_nt120", "fresnel_zout_n25_nt70", "brdf_n13_nt97", "aer_n9_nt600",
                "brdf_zout_n30_nt110", "aer_n32_nt75", "aer_n26_nt90"]


@pytest.mark.parametrize("mode", ["one_workgroup_per_bin", "persistent_order_scheduled", "one_order_per_launch", "three_orders_per_launch"])
@pytest.mark.parametrize("name", STREAM_CASES)
def test_streamed_solver_launch_forms_vs_golden(gpu_pkg, monkeypatch, name, mode):
    """The streamed solver runs few bins in its order-parallel form (the tests above: their batches are small) and many bins
    with one workgroup per bin; that form forced for the small cases, ONE persistent launch with order-scheduled tasks from
    per-XCD queues, and order-synchronous launches give the same records against the reference."""
    import os
    if mode == "one_workgroup_per_bin":
        monkeypatch.setenv("SOSGPU_STREAM_SPEC", "0")
    elif mode == "persistent_order_scheduled":
        monkeypatch.setenv("SOSGPU_STREAM_PERSIST", "1")
    else:
        monkeypatch.setenv("SOSGPU_STREAM_ORDERS_PER_LAUNCH", "1" if mode == "one_order_per_launch" else "3")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sos_os_%s.npz" % name))
    case = cases.make_case(name)
    got = cases.run_gpu(gpu_pkg, case)
    for b, r in enumerate(got):
        assert np.array_equal(r["ig_counts"], g["ig%d" % b]), (name, b)
        cases.compare_records(r["records"], g["rec%d" % b], 1e-9, "%s bin %d" % (name, b))
        assert np.allclose([r["emoins"], r["eplus"]], g["flux%d" % b], rtol=1e-9, atol=0)


def test_streamed_persistent_many_bins_bitwise_equal_to_per_bin_launch(gpu_pkg, monkeypatch):
    """More bins than the chip hosts workgroups, ragged level counts, a malformed bin in the middle: the persistent
    order-scheduled launch returns bit for bit what the one-workgroup-per-bin launch returns."""
    import torch
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(12, 35.0)
    al, be, ga, ze = S.hg_phase(24, 0.7)
    nb = 1500
    rng = np.random.default_rng(3)
    b = S.ckd_bins(nb, 140, seed=9)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, 24)
    nt = rng.integers(66, 141, nb).astype(np.int32)
    nt[700] = 0                                            # malformed: flagged norders = -1, the others unaffected
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=24, ro=0.2)
    bins = cx.upload_bins(h, x, y, nt=nt, iborm=rng.integers(0, 25, nb).astype(np.int32))
    monkeypatch.setenv("SOSGPU_STREAM_PERSIST", "1")
    monkeypatch.setenv("SOSGPU_STREAM_QTAIL", "16")        # order-by-order hand-overs until 16 bins per queue are left
    out_p = cx.solve(bins)
    torch.cuda.synchronize()
    monkeypatch.setenv("SOSGPU_STREAM_PERSIST", "0")
    out_b = cx.solve(bins)
    torch.cuda.synchronize()
    assert int(out_p["norders"][700]) == -1 and int((out_p["norders"] > 0).sum()) == nb - 1
    for k in ("norders", "iglast", "flux"):
        assert torch.equal(out_p[k], out_b[k]), k
    no = out_p["norders"].cpu().numpy()
    rp, rb = out_p["rec"].cpu().numpy(), out_b["rec"].cpu().numpy()
    for i in range(nb):
        assert np.array_equal(rp[i, :max(no[i], 0)], rb[i, :max(no[i], 0)]), i
    cx.close()


# every (NW, RTWH, KHT, ZO, SURF) instantiation of k_sos_stream -- N = 13: <4,1,4>, 25: <4,2,5>, 30: <4,2,6>, 41: <4,2,8>, 49: <8,2,16> -- with K forced so that
# a bin needs several rounds of order tasks + replay (the hand-over of the I3 terms must be checked in every variant)
_ALL_VARIANTS = [(3, n, 24, surf, zout, 4) for n in (13, 25, 30, 41, 49) for surf in (False, True) for zout in (False, True)]


@pytest.mark.parametrize("nb,n,os_nb,surf,zout,k", [(1, 13, 24, False, False, 0), (37, 13, 24, False, False, 0),
                                                    (100, 13, 24, True, False, 0), (1, 25, 80, True, True, 0),
                                                    (3, 25, 80, True, True, 5), (2, 41, 48, False, True, 7)] + _ALL_VARIANTS)
def test_streamed_order_parallel_form_equals_per_bin_launch(gpu_pkg, monkeypatch, nb, n, os_nb, surf, zout, k):
    """Few bins: the Fourier orders of a bin run as independent workgroups and the stop tests are replayed afterwards (one
    round of all orders for 1 bin, several rounds of K orders for 100 bins or with K forced).  Records of the orders a bin ran,
    order counts, scattering-order counts and fluxes are bit for bit those of the one-workgroup-per-bin launch; ragged level
    counts, per-bin IBORM, a malformed bin, output levels with and without surface matrices."""
    import torch
    import cases
    S = gpu_pkg.synth
    mu, w, n0 = S.gauss_angles(n - 1, 35.0)
    al, be, ga, ze = S.hg_phase(os_nb, 0.7)
    rng = np.random.default_rng(17)
    b = S.ckd_bins(nb, 150 if not zout else 97, seed=21)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, os_nb)
    kw = dict(ro=0.2, ifresnel=0 if surf else 1)
    if surf:
        kw.update(imat_surf=1, rsurf=cases._surf_matrices(n, os_nb, 5))
    cx = gpu_pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=os_nb, **kw)
    if zout:                                               # full-length level grids (the output level must exist in every bin)
        bins = cx.upload_bins(h, x, y, zout=1.5, zprof=b["zprof"])
        nbad = 0
    else:
        nt = rng.integers(66, 151, nb).astype(np.int32)
        nbad = 1 if nb > 2 else 0
        if nbad:
            nt[nb // 2] = 0
        bins = cx.upload_bins(h, x, y, nt=nt, iborm=rng.integers(0, os_nb + 1, nb).astype(np.int32))
    if k:
        monkeypatch.setenv("SOSGPU_STREAM_SPEC_K", str(k))
    out_s = cx.solve(bins)
    torch.cuda.synchronize()
    monkeypatch.setenv("SOSGPU_STREAM_SPEC", "0")
    out_b = cx.solve(bins)
    torch.cuda.synchronize()
    for key in ("norders", "iglast", "flux"):
        assert torch.equal(out_s[key], out_b[key]), key
    no = out_b["norders"].cpu().numpy()
    assert (no > 0).sum() == nb - nbad
    rs, rb = out_s["rec"].cpu().numpy(), out_b["rec"].cpu().numpy()
    for i in range(nb):
        assert np.array_equal(rs[i, :max(no[i], 0)], rb[i, :max(no[i], 0)]), i
    # include/sosgpu.h: only orders 0 .. norders-1 hold records -- the zero-filled buffer of solve() stays zero beyond them in
    # BOTH forms (the order-parallel form computes orders past the stop and clears their rows again)
    assert np.array_equal(rs, rb)
    for i in range(nb):
        assert not rs[i, max(no[i], 0):].any(), i
    cx.close()
