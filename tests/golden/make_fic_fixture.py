#!/usr/bin/env python3
"""Trimmed copies of the reference's CKD data files for the wavelengths the tests use (authoring container only).

tests/golden/fic/ mirrors $SOS_ABS_ROOT/fic: SO2-NO2 and the aerosol component tables (Data_*, IRefrac_*) as they are, and for every gas the 50-interval coefficient file holding
each test wavenumber with only THAT spectral interval's block kept -- every other interval is written as "no absorption"
(NMAXAI = 0, which the file format provides for, SOS_SUB_TRS.F:745-757).  Header, temperature / pressure / mole-fraction
grids and the kept blocks are copied byte for byte, so reading the kept interval gives exactly what the full file gives
(tests/test_absorption.py checks that where /root/reference is present).  Data only; 1.4 GB -> < 1 MB."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

A = importlib.import_module("radiativetransfer-sos_amd.absorption")
SRC = "/root/reference/fic"
DST = os.path.join(HERE, "fic")
NUS = [1e4 / 0.762, 15925.0]          # O2-A band (5 bins), H2O x O2 (25 bins)
NUSTEP = 10.0


def trim(nabs, nus):
    rel, numax, numin = A.ckd_file_name(nabs, nus[0], NUSTEP)
    lines = open(os.path.join(SRC, rel)).read().split("\n")
    nhead = 21 if nabs == 1 else 18
    out = lines[:nhead + 5 + (2 if nabs == 1 else 0)]
    pos = len(out)
    nt = int(lines[nhead + 1].split()[0])
    npr = int(lines[nhead + 3].split()[0])
    nc = int(lines[nhead + 5].split()[0]) if nabs == 1 else 1
    nwa = int((numax - numin) / NUSTEP)
    for iwa in range(nwa):
        head = lines[pos]
        f = head.split()
        nmax = int(f[5])
        lo, hi = float(f[4]), float(f[3])
        nrow = 0 if nmax == 0 else 1 + nmax * nc * npr
        if nmax and any(lo <= nu <= hi for nu in nus):
            out += lines[pos:pos + 1 + nrow]
        else:
            out.append(head[:head.rstrip().rfind(" ") + 1] + "0" if nmax else head)
        pos += 1 + nrow
    dst = os.path.join(DST, rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as g:
        g.write("\n".join(out) + "\n")
    return dst


if __name__ == "__main__":
    shutil.rmtree(DST, ignore_errors=True)
    os.makedirs(DST)
    shutil.copy(os.path.join(SRC, "SO2-NO2"), os.path.join(DST, "SO2-NO2"))
    # component tables of the WMO and Shettle & Fenn aerosol models (7 small text files, as they are)
    for name in sorted(os.listdir(SRC)):
        if name.startswith("Data_") or name.startswith("IRefrac_"):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, name))
    by_file = {}
    for nu in NUS:
        by_file.setdefault(A.ckd_file_name(1, nu, NUSTEP)[1], []).append(nu)
    tot = 0
    for nus in by_file.values():
        for nabs in range(1, 9):
            d = trim(nabs, nus)
            tot += os.path.getsize(d)
    print("fixture tree", DST, tot, "bytes")
