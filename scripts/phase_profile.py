"""Diagnostic: per-phase cycle breakdown of k_sos_os (needs a library built with -DSOS_PROFILE_PHASES,
selected through SOSGPU_LIB)."""
import sys, os, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
bins = S.ckd_bins(nb, nt, seed=1234)
h, x, y, iborm = S.rescale_profile(bins["h"], bins["xdel"], bins["ydel"], 0.0, 0.95, 0.95, 80)
cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, ro=0.1)
b = cx.upload_bins(h, x, y)
ph = torch.zeros((nb, 8), dtype=torch.int64, device=cx.device)
pkg.capi.check(pkg.capi.lib().sosgpu_debug_phase_buffer(cx._h, C.c_void_p(ph.data_ptr())), "phase")
out = cx.solve(b)
torch.cuda.synchronize()
p = ph.cpu().numpy().astype(np.float64) / 4.0     # 4 waves add their own cycles
names = ["order1 fill", "scan", "gemm", "writeback", "reduce/tests", "ground_bc", "fourier", "unused"]
tot = p.sum()
steps = (out["iglast"].cpu().numpy().clip(min=1) - 1).sum()
print("bins", nb, "ig steps", steps, "kernel ms", cx.last_solve_ms())
for k, nme in enumerate(names):
    print("%-14s %6.2f %%   %8.3f us per ig-step (100 MHz s_memrealtime)" % (nme, 100 * p[:, k].sum() / tot, p[:, k].sum() / steps / 100.0))
