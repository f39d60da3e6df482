"""Host cost of the per-wavelength context calls (sosgpu_create, sosgpu_set_surface_matrices_async, sosgpu_noyaux,
sosgpu_destroy) at the hyperspectral configuration (16 Gauss angles, OS_NB = 32, surface matrices), on an idle GPU and with a
long kernel running on another stream.  Usage: python scripts/ctx_latency.py [count]"""
import ctypes as C, gc, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
pkg = importlib.import_module("radiativetransfer-sos_amd")
S, capi = pkg.synth, pkg.capi
L = capi.lib()
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, os_nb = 17, 32
mu, w, n0 = S.gauss_angles(n - 1, 35.0)
al, be, ga, ze = S.hg_phase(os_nb, 0.7)
rs = torch.from_numpy(np.ascontiguousarray(cases._surf_matrices(n, os_nb, 5), dtype=np.float32)).cuda()
dp = lambda a: a.ctypes.data_as(C.c_void_p)
wv = capi.Wave(n=n, os_nb=os_nb, n0=n0, imat_surf=1, ifresnel=0, ipolar=1, igmax=100, reserved=0, ro=0.1, ind_surf=1.34, ron=0.0279)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(tag, background):
    side = torch.cuda.Stream()
    big = torch.zeros(1 << 26, device="cuda")
    t = dict(create=0.0, surface=0.0, noyaux=0.0, destroy=0.0)
    hs = []
    for k in range(cnt):
        if background and k % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    big.add_(1.0)
        h = C.c_void_p()
        t0 = time.perf_counter()
        capi.check(L.sosgpu_create(C.byref(h), 0, C.byref(wv), dp(mu), dp(w), dp(al), dp(be), dp(ga), dp(ze), os_nb), "create")
        t1 = time.perf_counter()
        capi.check(L.sosgpu_set_surface_matrices_async(h, C.c_void_p(rs.data_ptr()), st), "surface")
        t2 = time.perf_counter()
        capi.check(L.sosgpu_noyaux(h, st), "noyaux")
        t3 = time.perf_counter()
        t["create"] += t1 - t0; t["surface"] += t2 - t1; t["noyaux"] += t3 - t2
        hs.append(h)
        if len(hs) == 256:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for q in hs:
                L.sosgpu_destroy(q)
            t["destroy"] += time.perf_counter() - t0
            hs = []
    torch.cuda.synchronize()
    for q in hs:
        L.sosgpu_destroy(q)
    print("%-28s" % tag, "  ".join("%s %.1f us" % (k, 1e6 * v / cnt) for k, v in t.items()), flush=True)


run("warm-up (pool fill)", False)
run("idle GPU", False)
run("kernels on another stream", True)
gc.disable()
run("idle GPU, gc disabled", False)
