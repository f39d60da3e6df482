"""Latency of ONE solve of a small band (1 / 5 / 25 / 100 bins, N = 41, OS_NB = 80, real level grids) in the streamed solver's
launch forms: order-parallel (default for few bins) against one workgroup per bin.  Usage: python scripts/latency_bench.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
for nb in (1, 5, 25, 100):
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.1)
    alt, tabs = bench.realistic_columns(nb)
    bins = cx.make_profiles(nb, 0.0948, 8.0, 0.3, 2.0, alt, tabs, piz=0.95, piztr=0.95)
    out = cx.alloc_outputs(nb)
    res = {}
    for mode in ("1", "0"):
        os.environ["SOSGPU_STREAM_SPEC"] = mode
        t0 = time.perf_counter(); cx.solve(bins, out); torch.cuda.synchronize(); first = time.perf_counter() - t0
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); cx.solve(bins, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[mode] = (first, min(ts), int(out["norders"].max()))
    print("%3d bins (NT %d..%d): order-parallel %.2f ms (first call %.2f ms), one workgroup per bin %.2f ms (first %.2f); orders %d"
          % (nb, int(bins["nt"].min()), int(bins["nt"].max()), res["1"][1] * 1e3, res["1"][0] * 1e3, res["0"][1] * 1e3,
             res["0"][0] * 1e3, res["1"][2]), flush=True)
    cx.close()
