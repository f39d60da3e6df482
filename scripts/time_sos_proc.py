"""Wall-clock of run_sos.sos_proc per call (host + device) for the golden parameter sets: what a hyperspectral loop pays per
wavelength.  Usage (GPU box): python scripts/time_sos_proc.py [case ...]"""
import cProfile
import importlib
import json
import os
import pstats
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("SOS_ABS_ROOT", GOLD)
pkg = importlib.import_module("radiativetransfer-sos_amd")
rs = pkg.run_sos
import torch

names = sys.argv[1:] or ["sos_proc_cfg5_ckd_maignan_25bins", "sos_proc_ckd_h2o_o2_25bins_flatsea", "sos_proc_cfg2_lnd_lambert",
                         "sos_proc_cfg4_glitter_bilnd"]
for name in names:
    g = np.load(os.path.join(GOLD, name + ".npz"))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT", "-SOS.Trans": "NO_OUTPUT"})
    kw = rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False)
    rs.sos_proc(**kw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); rs.sos_proc(**kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-40s %.3f s per call (min of 3: %.3f)" % (name, float(np.mean(ts)), min(ts)), flush=True)
    if name == names[0]:
        pr = cProfile.Profile(); pr.enable(); rs.sos_proc(**kw); torch.cuda.synchronize(); pr.disable()
        st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)


# a spectrum: the first case at 48 wavelengths around its own, sequentially and through sos_proc_many
g = np.load(os.path.join(GOLD, names[0] + ".npz"))
user = json.loads(str(g["user_json"]))
user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT", "-SOS.Trans": "NO_OUTPUT", "-SOS_Main.ResRoot": ""})
kws = []
for i in range(48):
    u = dict(user)
    u["-ANG.Thetas"] = float(user["-ANG.Thetas"]) + 0.01 * i          # (the trimmed CKD fixtures hold two wavenumbers only:
    kws.append(rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), u), trace=False))   # vary the geometry instead)
t0 = time.perf_counter()
seq = [rs.sos_proc(**kw) for kw in kws]
t_seq = time.perf_counter() - t0
for nw in (4, 8, 16):
    t0 = time.perf_counter()
    par = rs.sos_proc_many(kws, n_workers=nw)
    t_par = time.perf_counter() - t0
    same = all(np.array_equal(np.asarray(a), np.asarray(b)) for x, y in zip(seq, par) for a, b in zip(x, y))
    print("48 calls of %s: sequential %.1f calls/s, sos_proc_many(%d workers) %.1f calls/s, identical: %s"
          % (names[0], 48 / t_seq, nw, 48 / t_par, same), flush=True)
