"""Worker of tests/test_spectrum.py::test_sos_spectrum_wavelengths_over_ranks_on_one_gpu: run_sos.sos_spectrum under
torch.distributed (several ranks on cuda:0, gloo): the wavelengths are partitioned over the ranks, results gathered."""
import argparse
import hashlib
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
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["SOS_ABS_ROOT"] = os.path.join(ROOT, "tests", "golden")
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    rs = pkg.run_sos
    import cases
    import spectrum_cases
    names = ["ckd_h2o_o2_25bins_flatsea", "cfg1_lambert", "ckd_o2a_5bins", "cfg2_lnd_lambert", "rand_00", "rand_13", "rand_05",
             "cfg5_ckd_maignan_25bins", "flatsea_zout", "rand_17", "land_roujean"]
    tmp = os.path.join(os.path.dirname(a.out), "rank%d" % rank)
    kws, golds, coefs, rtols = spectrum_cases.build(rs, tmp, names=names)
    mine = [int(i) for i in pkg.dist.balanced_shards(rs.spectrum_costs(kws), world)[rank]]
    outs = rs.sos_spectrum(kws)
    torch.cuda.synchronize()
    assert all(o is not None for o in outs)
    for out, g, coef, rtol in zip(outs, golds, coefs, rtols):
        cases.compare_proc_outputs(rs, out, g, coef_tronca=coef, rtol=rtol)
    h = hashlib.sha256()
    for out in outs:
        for x in out:
            h.update(np.ascontiguousarray(np.asarray(x, dtype=np.float64)).tobytes())
    part = rs.sos_spectrum(kws, gather=False)              # without the gather: only this rank's wavelengths are filled
    assert [i for i, o in enumerate(part) if o is not None] == mine
    dig, own = [None] * world, [None] * world
    dist.all_gather_object(dig, h.hexdigest())
    dist.all_gather_object(own, mine)
    if rank == 0:
        with open(a.out, "w") as f:
            json.dump({"world": world, "n": len(kws), "digests": dig, "owners": own}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
