#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ from the REAL reference Fortran
(oracle/_ref/libsos_ref.so, built from /root/reference by oracle/Makefile with amdflang).

Run in the authoring container only:  python tests/golden/make_golden.py
The fixtures hold inputs and expected outputs (data only); the reference itself never travels.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import ref_ctypes as R  # noqa: E402


def gen_sos_os():
    for name in cases.ALL_CASES:
        case = cases.make_case(name)
        d = dict(rmu=case["rmu"], ga=case["ga"], n0=case["n0"], os_nb=case["os_nb"], iborm=case["iborm"],
                 alpha=case["coefs"][0], beta=case["coefs"][1], gamma=case["coefs"][2], zeta=case["coefs"][3],
                 nbins=len(case["bins"]))
        for b, (h, x, y, z) in enumerate(case["bins"]):
            r = cases.run_cpu(R, case, b)
            assert r["ier"] == 0
            d["h%d" % b], d["xdel%d" % b], d["ydel%d" % b], d["zprof%d" % b] = h, x, y, z
            d["rec%d" % b] = r["records"]
            d["ig%d" % b] = r["ig_counts"]
            d["flux%d" % b] = np.array([r["emoins"], r["eplus"]])
            print(name, b, "F", len(r["records"]), "ig", list(r["ig_counts"][:6]))
        np.savez_compressed(os.path.join(HERE, "sos_os_%s.npz" % name), **d)


def gen_noyaux():
    S = cases.S
    mu, w, n0 = S.gauss_angles(12, 35.0)
    al, be, ga, ze = S.hg_phase(24, 0.6)
    d = dict(mu=mu, n0=n0, alpha=al, beta=be, gamma=ga, zeta=ze)
    for is_ in (0, 1, 2, 3, 12, 24):
        k = R.sos_noyaux(is_, -mu[n0 - 1], mu, 24, al, be, ga, ze)
        for key, v in k.items():
            d["is%d_%s" % (is_, key)] = v
    np.savez_compressed(os.path.join(HERE, "noyaux_n13.npz"), **d)
    print("noyaux ok")


if __name__ == "__main__":
    gen_noyaux()
    gen_sos_os()
