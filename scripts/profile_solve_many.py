"""Host-side profile of solver.solve_many (where do the ~0.25 ms per wavelength go?)."""
import cProfile, importlib, os, pstats, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
mu, w, n0 = S.gauss_angles(40, 35.0)
items = []
for i in range(64):
    al, be, ga, ze = S.hg_phase(80, 0.75)
    b = S.ckd_bins(32, 30, seed=100 + i)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, 80)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.1)
    bins = cx.upload_bins(h, x, y)
    items.append((cx, bins, torch.from_numpy(b["aik"]).to(cx.device)))
pkg.solver.solve_many(items, 16); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    pkg.solver.solve_many(items, 16)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
