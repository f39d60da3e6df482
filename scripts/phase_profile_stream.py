"""Diagnostic: per-phase time breakdown of k_sos_stream on the realistic level-grid mix (needs a library built with
-DSOS_PROFILE_PHASES, selected through SOSGPU_LIB; radiativetransfer-sos_amd/build_ext.py build(extra=..., out=...))."""
import sys, os, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.1)
alt, tabs = bench.realistic_columns(nb)
order = np.argsort(-tabs[:, -1], kind="stable")
bins = cx.make_profiles(nb, 0.0948, 8.0, 0.3, 2.0, alt, tabs[order], piz=0.95, piztr=0.95)
ph = torch.zeros((nb, 8), dtype=torch.int64, device=cx.device)
pkg.capi.check(pkg.capi.lib().sosgpu_debug_phase_buffer(cx._h, C.c_void_p(ph.data_ptr())), "phase")
out = cx.solve(bins)
torch.cuda.synchronize()
p = ph.cpu().numpy().astype(np.float64) / float(os.environ.get("PHASE_WAVES", "4"))     # the waves add their own stamps (2 for the SOS_PHASE_ROLE builds)
nt = bins["nt"].cpu().numpy()
igl = out["iglast"].cpu().numpy()
nch = (nt + 32) // 32
chunks = (np.clip(igl - 1, 0, None).sum(axis=1) * nch).sum()
names = ["stage wait", "fix-up", "gemm", "write-back", "sweeps", "store", "pass end/tests", "order-1 passes"]
tot = p.sum()
print("bins", nb, "chunk-contractions", chunks, "kernel ms", cx.last_solve_ms())
for k, nme in enumerate(names):
    print("%-16s %6.2f %%   %8.3f us per chunk-contraction (100 MHz s_memrealtime)" % (nme, 100 * p[:, k].sum() / tot, p[:, k].sum() / chunks / 100.0))
