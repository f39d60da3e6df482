"""the order-parallel form of k_sos_stream<4,2,ZO,SURF> handed the replay wrong I3 terms when its order
loop was left by a `break` right after the spec_i3 store.  This probe runs the witness configuration
(test_streamed_order_parallel_form_equals_per_bin_launch[1-25-80-True-True-0]) and saves, for the library selected by
SOSGPU_LIB, the I3 hand-over block (per Fourier order and thread), the per-bin-launch results and the order-parallel results."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb, n, os_nb = 1, 25, 80
mu, w, n0 = S.gauss_angles(n - 1, 35.0)
al, be, ga, ze = S.hg_phase(os_nb, 0.7)
b = S.ckd_bins(nb, 97, seed=21)
h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, os_nb)
cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=os_nb, ro=0.2, ifresnel=0, imat_surf=1, rsurf=cases._surf_matrices(n, os_nb, 5))
bins = cx.upload_bins(h, x, y, zout=1.5, zprof=b["zprof"])
os.environ["SOSGPU_STREAM_SPEC_K"] = "81"                      # every order in the first round: the block holds them all
out_s = cx.solve(bins)
torch.cuda.synchronize()
p, nd, off = C.c_void_p(), C.c_size_t(), C.c_size_t()
pkg.capi.check(pkg.capi.lib().sosgpu_debug_scratch(cx._h, C.byref(p), C.byref(nd), C.byref(off)), "debug_scratch")
nth = 256
cnt = nb * (os_nb + 1) * nth
i3 = np.zeros(cnt)
assert off.value and off.value + cnt <= nd.value
torch.cuda.synchronize()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipMemcpy(i3.ctypes.data_as(C.c_void_p), C.c_void_p(p.value + 8 * off.value), C.c_size_t(8 * cnt), 2)
assert rc == 0, rc
os.environ["SOSGPU_STREAM_SPEC"] = "0"
out_b = cx.solve(bins)
torch.cuda.synchronize()
tag = sys.argv[1]
np.savez(os.path.join(ROOT, "gpurun_out", "spec_probe_%s.npz" % tag), i3=i3.reshape(os_nb + 1, nth),
         nord_s=out_s["norders"].cpu().numpy(), nord_b=out_b["norders"].cpu().numpy(), igl_s=out_s["iglast"].cpu().numpy(),
         igl_b=out_b["iglast"].cpu().numpy(), rec_s=out_s["rec"].cpu().numpy(), rec_b=out_b["rec"].cpu().numpy())
print(tag, "norders spec", out_s["norders"].cpu().numpy(), "per-bin", out_b["norders"].cpu().numpy())
