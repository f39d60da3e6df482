"""Timing of sosgpu_profile (SOS_PROFILE on the device) for a batch of CKD bins with random gas columns."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cases
pkg = importlib.import_module("radiativetransfer-sos_amd")
S = pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mu, w, n0 = S.gauss_angles(40, 35.0)
al, be, ga, ze = S.hg_phase(80, 0.75)
cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.1)
c = cases.profile_case("gas_weak")
rng = np.random.default_rng(5)
scale = np.exp(rng.uniform(np.log(1e-3), np.log(30.0), nb))          # k_b log-uniform in [1e-3, 30]
tabs = scale[:, None] * c["tabs"][None, :] / 0.4
p = cx.make_profiles(nb, c["tr"], c["hr"], c["ta"], c["ha"], c["altabs"], tabs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    p = cx.make_profiles(nb, c["tr"], c["hr"], c["ta"], c["ha"], c["altabs"], tabs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
nt = p["nt"].cpu().numpy()
print("bins %d  %.3f ms per batch (incl. H2D of tabs and the host no-gas step)  %.0f profiles/s  NT min/mean/max %d/%.0f/%d  failed %d"
      % (nb, ms, nb / ms * 1e3, nt[nt > 0].min(), nt[nt > 0].mean(), nt.max(), int((nt < 0).sum())))
out = cx.solve(p)
torch.cuda.synchronize()
print("solve of the same bins: %.1f ms (field-in-HBM variants)" % cx.last_solve_ms())
