"""Hyperspectral shape of the workload (BASELINE config 5): MANY wavelengths with FEW CKD bins each.  Every wavelength has its
own context (its own source operators); the bins of one wavelength fill only a fraction of the chip, so wavelengths are put on
several HIP streams and overlap (solver.solve_many) -- or all their bins go into ONE launch with a per-bin context
(solver.solve_spectrum).  Prints bins/s for W wavelengths x B bins over S streams (same per-bin work as bench.py's
headline: N = 41, NT = 30, OS_NB = 80).  Usage: python scripts/spectrum_bench.py [W] [B] [S ...]"""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # HIP multiplexes streams onto this many hardware queues (default 4)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
streams_list = [int(x) for x in sys.argv[3:]] or [1, 4, 16]

mu, w, n0 = S.gauss_angles(40, 35.0)
rng = np.random.default_rng(11)
ctxs = []
for i in range(W):
    g = 0.70 + 0.1 * rng.random()                      # a different phase function per wavelength
    al, be, ga, ze = S.hg_phase(80, g)
    b = S.ckd_bins(B, 30, seed=100 + i)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, 80)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.1)
    bins = cx.upload_bins(h, x, y, order="cost")
    aik = torch.from_numpy(b["aik"][bins["perm"]]).to(cx.device)
    ctxs.append((cx, bins, aik, cx.alloc_outputs(B)))
torch.cuda.synchronize()
for ns in streams_list:
    items = [(cx, bins, aik) for cx, bins, aik, _ in ctxs]

    def run():
        pkg.solver.solve_many(items, n_streams=ns)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%3d wavelengths x %4d bins, %2d streams: %8.0f bins/s  (%.2f ms per spectrum, %.3f ms per wavelength)"
          % (W, B, ns, W * B / dt, dt * 1e3, dt * 1e3 / W), flush=True)
# one launch for the whole spectrum: per-bin context table
table = pkg.solver.ContextTable([c[0] for c in ctxs])
bins_all, cob, seg = pkg.solver.concat_bins([c[1] for c in ctxs])
aik_all = torch.cat([c[2] for c in ctxs])
out_all = ctxs[0][0].alloc_outputs(bins_all["nb"], zero=False)
ref = pkg.solver.solve_many([(cx, bins, aik) for cx, bins, aik, _ in ctxs], n_streams=4)
got = pkg.solver.solve_spectrum(table, bins_all, cob, seg, aik_all, out=out_all)
torch.cuda.synchronize()
assert all(torch.equal(got[0][i], ref[i][0][0]) for i in range(W)), "one-launch spectrum differs from the per-wavelength calls"
t0 = time.perf_counter()
for _ in range(3):
    pkg.solver.solve_spectrum(table, bins_all, cob, seg, aik_all, out=out_all)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("%3d wavelengths x %4d bins, ONE launch: %8.0f bins/s  (%.2f ms per spectrum; kernel %.2f ms)"
      % (W, B, W * B / dt, dt * 1e3, ctxs[0][0].last_solve_ms()), flush=True)
for cx, *_ in ctxs:
    cx.close()
