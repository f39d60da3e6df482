"""A/B of the two context-passing forms on bench.py's headline workload (N = 41, NT = 30, OS_NB = 80, one wavelength): the
by-value kernel argument (sosgpu_os_solve) against the per-bin context table with a single entry (sosgpu_os_solve_multi).
Usage: python scripts/multi_headline.py [bins] [nt]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("radiativetransfer-sos_amd")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wl = bench.build_workload(pkg.synth, nb, nt, 1234, 0.75)
cx = pkg.SosContext(wl["mu"], wl["w"], wl["n0"], *wl["coefs"], iborm_max=wl["iborm"], ro=0.1)
bins = cx.upload_bins(wl["h"], wl["xdel"], wl["ydel"], order="cost")
out = cx.alloc_outputs(nb)
out2 = cx.alloc_outputs(nb)
table = pkg.solver.ContextTable([cx])
cob = torch.zeros(nb, dtype=torch.int32, device=cx.device)
seg = torch.tensor([0, nb], dtype=torch.int32, device=cx.device)
aik = torch.from_numpy(wl["aik"][bins["perm"]]).to(cx.device)


def single():
    cx.solve(bins, out)


def multi():
    pkg.solver.solve_spectrum(table, bins, cob, seg, aik, out=out2)


for name, fn in (("kernel argument", single), ("context table ", multi), ("kernel argument", single), ("context table ", multi)):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        fn()
        ms.append(cx.last_solve_ms())
    print("%s: %8.0f bins/s  (kernel %.3f ms, min of 5)" % (name, nb / min(ms) * 1e3, min(ms)), flush=True)
assert all(torch.equal(out[k], out2[k]) for k in ("rec", "norders", "iglast", "flux"))
print("bit-identical")
cx.close()
