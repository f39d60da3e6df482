"""One invocation of every kernel either side of the solver at the BASELINE config-4 / config-5 sizes (N = 41, OS_NB = 80),
for the rocprofv3 --kernel-trace --stats pass of scripts/collect_profiles.sh: noyaux / pack kernels, Cox-Munk glitter
matrices, land-surface matrices (Roujean + Maignan), Mie records, SOS_ABSPROFILE + SOS_PROFILE of 4096 bins, SOS_TRPHI."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

pkg = importlib.import_module("radiativetransfer-sos_amd")
S, surf, A = pkg.synth, pkg.surface, pkg.aerosols
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
for rep in range(3):
    rs = surf.glitter_matrices(mu, w, 7.0, 1.34, 80, 80, 160)["rsurf"]
    land = surf.land_matrices(surf.land_model(7, 0.2, 0.03, 0.25, coef_c=4.0), mu, w, 1.5, 80, 80, 160)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.0, imat_surf=1, rsurf=rs)
    xmu, xhr = A.mie_angles(40)
    rec = A.mie_records(xmu, 1.45, -0.003, 0.0001, 200.0)
    alt, tabs = bench.realistic_columns(4096)
    ik = np.ones((4096, 8), dtype=np.int32)
    xk = np.abs(np.random.default_rng(0).normal(size=(8, 5, 49))) * 1e-24
    ro = np.full((8, 49), 1e21)
    t2 = cx.absorption_profiles(ik, xk, ro)
    bins = cx.make_profiles(4096, 0.0948, 8.0, 0.3, 2.0, alt, tabs, piz=0.95, piztr=0.95)
    out = cx.solve(cx.upload_bins(*[x[:64] for x in S.rescale_profile(*S.profile(30, k_abs=np.linspace(0, 3, 64))[:3], 0.0, 0.95, 0.95, 80)[:3]]))
    rec_a, sc = cx.aggregate(out, np.full(64, 1 / 64.))
    tp = cx.trphi(rec_a[0], int(out["norders"].max()), 0.4, 0.0, np.radians(np.arange(0, 361, 1.0)), igli=1, wind=7.0)
    torch.cuda.synchronize()
    cx.close()
print("aux kernels ok", tuple(rs.shape), tuple(land.shape), len(rec["alpha"]), int(bins["nt"].max()), tuple(tp.shape))
