"""Headline workload at other direction counts (Gauss orders): the rate of the kernel variant that runs there and its fraction
of the FP64 matrix peak.  Usage: python scripts/n_sweep.py [nt] [ng ...]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ngs = [int(x) for x in sys.argv[2:]] or [12, 20, 24, 32, 40, 41, 48, 56, 64, 80]
for ng in ngs:
    mu, w, n0 = S.gauss_angles(ng, 35.0)
    os_nb = 80
    al, be, ga, ze = S.hg_phase(os_nb, 0.75)
    nb = 4096 if ng <= 41 else 2048
    b = S.ckd_bins(nb, nt, seed=1234)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, os_nb)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.1)
    bins = cx.upload_bins(h, x, y, order="cost")
    out = cx.alloc_outputs(nb)
    cx.solve(bins, out); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        cx.solve(bins, out)
        ms.append(cx.last_solve_ms())
    tot, exe = cx.solve_flops(bins, out)
    t = min(ms) * 1e-3
    steps = float(np.clip(out["iglast"].cpu().numpy() - 1, 0, None).sum()) / nb
    print("Gauss %2d  N=%2d  %8.0f bins/s  %6.2f ms per %d bins  %5.1f steps/bin  %5.1f TFLOP/s executed  frac %.3f"
          % (ng, len(mu), nb / t, t * 1e3, nb, steps, exe / t / 1e12, exe / t / 78.6e12), flush=True)
    cx.close()
