"""Worker of tests/test_dist_gpu.py::test_sos_proc_sharded: run_sos.sos_proc under torch.distributed (several ranks on cuda:0,
gloo reduce): every rank takes a slice of the CKD bins of the band and must return the reference's outputs."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["SOS_ABS_ROOT"] = os.path.join(ROOT, "tests", "golden")
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    rs = pkg.run_sos
    import cases
    g = np.load(os.path.join(ROOT, "tests", "golden", "sos_proc_%s.npz" % a.case))
    user = json.loads(str(g["user_json"]))
    tmp = os.path.dirname(a.out)
    user.update({"-SOS_Main.Log": "NO_LOG_FILE", "-SOS_Main.ResRoot": os.path.join(tmp, "res")})
    user.setdefault("-SOS.Flux", "NO_OUTPUT")
    coef = None
    if user["-AER.AOTref"] != 0.0:
        f = os.path.join(tmp, "Aerosols_user_%d.txt" % rank)
        rs.write_aerosols_file(f, {k: g["aer_" + k] for k in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")}, *g["kmat"])
        user["-AER.UserFile"] = f
        coef = 0.0
    out = rs.sos_proc(**rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
    torch.cuda.synchronize()
    cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=2e-7 if int(user.get("-SURF.Type", 0)) >= 3 else 1e-9)
    # every rank holds the same result; the files are written by rank 0 only
    allv = [None] * world
    dist.all_gather_object(allv, float(np.asarray(out[5]).sum()))
    files = sorted(os.listdir(os.path.join(tmp, "res", "SOS"))) if rank == 0 else []
    if rank == 0:
        with open(a.out, "w") as f:
            json.dump({"world": world, "sums": allv, "files": files}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
