"""Whole sos_proc calls in a plain sequential loop (32 calls differing in the solar angle; host + device, result files off): what a
per-wavelength user pays per call.  Usage (GPU box): python scripts/sos_proc_loop.py"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("SOS_ABS_ROOT", GOLD)
pkg = importlib.import_module("radiativetransfer-sos_amd")
rs = pkg.run_sos
import torch
def kws(name, n):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    user = json.loads(str(g["user_json"]))
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS.Flux": "NO_OUTPUT", "-SOS.Trans": "NO_OUTPUT", "-SOS_Main.ResRoot": ""})
    out = []
    for i in range(n):
        u = dict(user); u["-ANG.Thetas"] = float(user["-ANG.Thetas"]) + 0.01 * i
        out.append(rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), u), trace=False))
    return out
for name in ("sos_proc_cfg2_lnd_lambert", "sos_proc_cfg4_glitter_bilnd", "sos_proc_cfg5_ckd_maignan_25bins"):
    k = kws(name, 32)
    rs.sos_proc(**k[0]); torch.cuda.synchronize()
    ts = []
    for kw in k:
        t0 = time.perf_counter(); rs.sos_proc(**kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    seq = 1e3 / ts.mean()
    rs.sos_proc_many(k[:4]); torch.cuda.synchronize()
    t0 = time.perf_counter(); rs.sos_proc_many(k); torch.cuda.synchronize(); many = len(k) / (time.perf_counter() - t0)
    print("%-36s 32 different calls in a plain loop: median %.1f ms, min %.1f, max %.1f, %.1f calls/s;  sos_proc_many / sos_spectrum: %.1f calls/s (%.2f x)"
          % (name, np.median(ts), ts.min(), ts.max(), seq, many, many / seq), flush=True)
