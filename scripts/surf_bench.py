"""Timing of the BRDF-surface variant (IMAT_SURF = 1, synthetic matrices) against the Lambert variant, same bins."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
b = S.ckd_bins(nb, 30, seed=1234)
h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, 80)
for name, kw in (("lambert", dict(ro=0.1)), ("brdf", dict(ro=0.02, imat_surf=1, rsurf=cases._surf_matrices(len(mu), iborm, 7)))):
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, **kw)
    bins = cx.upload_bins(h, x, y, order="cost")
    ph = None
    if os.environ.get("SOSGPU_LIB", "").endswith("phase.so"):
        import ctypes as C
        ph = torch.zeros((nb, 8), dtype=torch.int64, device=cx.device)
        pkg.capi.check(pkg.capi.lib().sosgpu_debug_phase_buffer(cx._h, C.c_void_p(ph.data_ptr())), "phase")
    out = cx.solve(bins); torch.cuda.synchronize()
    if ph is not None:
        p = ph.cpu().numpy().astype(np.float64) / 4.0
        st = (out["iglast"].cpu().numpy().clip(min=1) - 1).sum()
        print(name, " ".join("%s %.2f" % (n_, p[:, k].sum() / st / 100.0) for k, n_ in enumerate(["o1", "scan", "gemm", "wb", "tests", "bc", "fourier"])), "us per ig-step")
        pkg.capi.check(pkg.capi.lib().sosgpu_debug_phase_buffer(cx._h, None), "phase")
    ts = []
    for _ in range(3):
        cx.solve(bins, out); ts.append(cx.last_solve_ms())
    steps = int((out["iglast"].cpu().numpy().clip(min=1) - 1).sum())
    print("%-8s %d bins: %.3f ms, %d ig-steps, %.2f us per 512-slot step" % (name, nb, np.mean(ts), steps, np.mean(ts) * 1e3 / (steps / 512)))
    cx.close()
