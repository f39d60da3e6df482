"""Throughput of the solver on the BASELINE.json configurations other than the headline one (2048 synthetic bins each):
cfg1 Rayleigh N=25, cfg2 headline, cfg3 flat sea (Fresnel), cfg4 Cox-Munk glitter matrices made by sosgpu_glitter."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048


def run(name, ng, g, kw, rayleigh=False, os_nb=80):
    mu, w, n0 = S.gauss_angles(ng, 35.0)
    al, be, ga, ze = S.hg_phase(os_nb, g)
    b = S.ckd_bins(nb, 30, seed=1234)
    x = np.zeros_like(b["xdel"]) if rayleigh else b["xdel"]
    h, x, y, iborm = S.rescale_profile(b["h"], x, b["ydel"], 0.0, 0.95, 0.95, os_nb)
    if kw.get("imat_surf"):
        kw = dict(kw, rsurf=pkg.surface.glitter_matrices(mu, w, 7.0, 1.34, iborm, os_nb, 2 * os_nb)["rsurf"][:iborm + 1])
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, **kw)
    bins = cx.upload_bins(h, x, y, order="cost")
    out = cx.solve(bins); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        cx.solve(bins, out); ts.append(cx.last_solve_ms())
    steps = int((out["iglast"].cpu().numpy().clip(min=1) - 1).sum())
    print("%-34s N=%2d  %8.0f bins/s  (%.2f ms per %d bins, %.0f Fourier orders, %d scattering steps per bin)"
          % (name, len(mu), nb / np.mean(ts) * 1e3, np.mean(ts), nb, float(out["norders"].float().mean()), steps // nb))
    cx.close()


run("cfg1 Rayleigh, Lambert", 24, 0.0, dict(ro=0.1), rayleigh=True)
run("cfg2 aerosol+Rayleigh, Lambert", 40, 0.75, dict(ro=0.1))
run("cfg3 flat sea (Fresnel)", 40, 0.75, dict(ro=0.02, ifresnel=1, ind_surf=1.34))
run("cfg4 Cox-Munk glitter 7 m/s", 40, 0.75, dict(ro=0.0, imat_surf=1))
