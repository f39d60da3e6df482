#!/usr/bin/env python3
"""Tabulate the six predefined atmospheres of the reference (-AP.AbsProfile.Type 1..6: tropical, mid-latitude summer /
winter, sub-arctic summer / winter, US standard 1962) by CALLING the compiled reference's DATATM (SOS_SUB_TRS.F:908)
through ctypes, and store the profile tables it returns as the product's data file

    radiativetransfer-sos_amd/data/afgl_atmospheres.npz :  donuser [6][50][11] float32
        columns: Z (km), P (hPa), T (K), H2O, CO2, O3, N2O, CO, CH4, O2 (ppmv), air density (cm-3); level 1 = ground

(the reference holds them as REAL*4 constants, so float32 keeps every bit; checked below).  Run in the authoring
container only (needs oracle/_ref/libsos_ref.so); the output is data, no reference source is copied."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_ctypes as R  # noqa: E402

out = np.zeros((6, 50, 11), dtype=np.float32)
for iatm in range(1, 7):
    d = R.datatm(iatm)["donuser"][:, :11]
    out[iatm - 1] = d.astype(np.float32)
    assert np.array_equal(out[iatm - 1].astype(np.float64), d), "atmosphere %d is not float32-exact" % iatm
path = os.path.join(ROOT, "radiativetransfer-sos_amd", "data", "afgl_atmospheres.npz")
np.savez_compressed(path, donuser=out)
print("wrote", path, os.path.getsize(path), "bytes; surface P/T:", out[:, 0, 1], out[:, 0, 2])
