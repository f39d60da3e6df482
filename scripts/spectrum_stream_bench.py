"""Hyperspectral shape with REAL level grids (streamed solver): W wavelengths x B bins through solver.solve_many, with the
order-parallel form of the streamed solver on (default for few bins) and off.  Usage: python scripts/spectrum_stream_bench.py [W] [B]"""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
mu, w, n0 = S.gauss_angles(40, 35.0)
rng = np.random.default_rng(3)
items = []
for i in range(W):
    al, be, ga, ze = S.hg_phase(80, 0.70 + 0.1 * rng.random())
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.1)
    alt, tabs = bench.realistic_columns(B, seed=100 + i)
    bins = cx.make_profiles(B, 0.0948, 8.0, 0.3, 2.0, alt, tabs, piz=0.95, piztr=0.95)
    aik = torch.full((B,), 1.0 / B, dtype=torch.float64, device=cx.device)
    items.append((cx, bins, aik))
torch.cuda.synchronize()
ref = None
for mode in ("1", "0", "1", "0"):
    os.environ["SOSGPU_STREAM_SPEC"] = mode
    got = pkg.solver.solve_many(items, n_streams=16); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        got = pkg.solver.solve_many(items, n_streams=16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    if ref is None:
        ref = got
    same = all(torch.equal(a[0], b[0]) for a, b in zip(got, ref))
    print("%d wavelengths x %d bins (NT 117..~400), 16 streams, order-parallel %s: %7.0f bins/s  (%.1f ms per spectrum)  same bands: %s"
          % (W, B, "on " if mode == "1" else "off", W * B / dt, dt * 1e3, same), flush=True)
# the whole spectrum in ONE launch (per-bin context table, one workgroup per bin) + one segmented aggregate
table = pkg.solver.ContextTable([it[0] for it in items])
bins_all, cob, seg = pkg.solver.concat_bins([it[1] for it in items])
aik_all = torch.cat([it[2] for it in items])
out_all = items[0][0].alloc_outputs(bins_all["nb"], zero=False)
got = pkg.solver.solve_spectrum(table, bins_all, cob, seg, aik_all, out=out_all); torch.cuda.synchronize()
same = all(torch.equal(got[0][i], ref[i][0][0]) for i in range(W))
t0 = time.perf_counter()
for _ in range(2):
    pkg.solver.solve_spectrum(table, bins_all, cob, seg, aik_all, out=out_all)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
print("%d wavelengths x %d bins, ONE launch (solve_spectrum):                  %7.0f bins/s  (%.1f ms per spectrum)  same bands: %s"
      % (W, B, W * B / dt, dt * 1e3, same), flush=True)
for cx, _, _ in items:
    cx.close()
