"""Developer aid (test infrastructure): run parity cases on the GPU and print per-order differences against the oracle.
usage: python tests/debug_compare.py <case> [...]   (case names of tests/cases.py, or x_<ng>_<nt>_<os_nb>_<g>_<zout>)"""
import sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import cases
from oracle import oracle_ctypes as O
pkg = importlib.import_module("radiativetransfer-sos_amd")
for name in sys.argv[1:]:
    case = cases.make_case(name)
    got = cases.run_gpu(pkg, case)
    for b, g in enumerate(got):
        ref = cases.run_cpu(O, case, b)
        print(name, b, "F", len(g["records"]), len(ref["records"]), "ig", list(g["ig_counts"]), list(ref["ig_counts"]))
        F = min(len(g["records"]), len(ref["records"]))
        a, r = g["records"][:F], ref["records"][:F]
        scale = np.abs(r[:, 0]).max()
        n = (a.shape[2] - 1) // 2
        for s in range(min(F, 6)):
            for c in range(3):
                e = np.abs(a[s, c] - r[s, c]) / scale
                print("  s", s, "c", c, "max err/scale %.2e" % e.max(), "at j", int(e.argmax()) - n, " up-max %.2e down-max %.2e" % (e[n+1:].max(), e[:n].max()))
        print("  emoins", g["emoins"], ref["emoins"], "eplus", g["eplus"], ref["eplus"])
