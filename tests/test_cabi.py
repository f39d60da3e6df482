"""CPU tests of the drop-in boundary: libsosgpu.so loads and exports every symbol include/sosgpu.h
declares (no compute without a GPU), argument validation, variant table."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "sosgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(sosgpu_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = pkg.capi.lib()
    for sym in declared:
        assert hasattr(L, sym), "libsosgpu.so does not export %s" % sym
    assert sorted(pkg.capi.EXPORTS) == declared


def test_version_and_errors(pkg):
    L = pkg.capi.lib()
    assert b"gfx950" in L.sosgpu_version()
    assert L.sosgpu_strerror(0) == b"ok"
    assert L.sosgpu_strerror(-3)


def test_create_rejects_bad_arguments(pkg):
    """Validation happens before any device work; without a GPU the only other outcome is NODEVICE."""
    L = pkg.capi.lib()
    S = pkg.synth
    mu, w, n0 = S.gauss_angles(8, 35.0)
    al, be, ga, ze = S.hg_phase(16, 0.5)
    h = C.c_void_p()
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    wv = pkg.capi.Wave(n=len(mu), os_nb=16, n0=0, imat_surf=0, ifresnel=0, ipolar=1, igmax=100, reserved=0,
                       ro=0.1, ind_surf=1.34, ron=0.0279)
    rc = L.sosgpu_create(C.byref(h), 0, C.byref(wv), dp(mu), dp(w), dp(al), dp(be), dp(ga), dp(ze), 16)
    assert rc == -1  # n0 must point at the solar direction
    wv.n0 = n0
    wv.n = 200
    rc = L.sosgpu_create(C.byref(h), 0, C.byref(wv), dp(mu), dp(w), dp(al), dp(be), dp(ga), dp(ze), 16)
    assert rc == -1
    wv.n = len(mu)
    rc = L.sosgpu_create(C.byref(h), 0, C.byref(wv), dp(mu), dp(w), dp(al), dp(be), dp(ga), dp(ze), 99)
    assert rc == -1  # iborm_max > os_nb


def test_product_path_has_no_cpu_fallback(pkg):
    """The host driver refuses to run without a GPU instead of silently computing on the CPU."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the refusal only exists on a box without one")
    S = pkg.synth
    mu, w, n0 = S.gauss_angles(8, 35.0)
    al, be, ga, ze = S.hg_phase(16, 0.5)
    try:
        pkg.SosContext(mu, w, n0, al, be, ga, ze)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("SosContext ran without a GPU")


def test_product_package_never_imports_oracle():
    pk = os.path.join(ROOT, "radiativetransfer-sos_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_ctypes" not in src and "ref_ctypes" not in src and "sos_oracle" not in src, f
