"""World-size-2 gloo test of the sharded bin aggregation (the N>1 path of bench.py without GPUs):
each rank forms the AIK-weighted partial sums of its shard, one all-reduce combines them, and the
result equals the serial SOS_AGGREGATE of the whole bin list within fp64 summation noise."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    nb, S1, W = 13, 6, 9
    rec = rng.normal(size=(nb, S1, 3, W))
    nf = rng.integers(1, S1 + 1, nb)
    for b in range(nb):
        rec[b, nf[b]:] = 0.0
    aik = rng.dirichlet(np.ones(nb))
    flux = rng.uniform(size=(nb, 2))
    scal = rng.uniform(size=(nb, 4))
    lo, hi = pkg.dist.shard_range(nb, rank, world)
    # local partial sums in the layout sosgpu_aggregate produces (include/sosgpu.h)
    prec = torch.from_numpy((aik[lo:hi, None, None, None] * rec[lo:hi]).sum(0, keepdims=True))
    a = aik[lo:hi]
    tdg = rng.uniform(size=(nb, 4))                       # TDIFMUG(1..N), N = 4
    pscal = torch.tensor([[(a * scal[lo:hi, 0]).sum(), (a * flux[lo:hi, 0]).sum(), (a * flux[lo:hi, 1]).sum(),
                           (a * np.exp(-scal[lo:hi, 1])).sum(), (a * np.exp(-scal[lo:hi, 2])).sum(),
                           (a * np.exp(-scal[lo:hi, 3])).sum(), a.sum(), float(nf[lo:hi].max()), -float(nf[lo:hi].min()),
                           0.0] + list((a[:, None] * tdg[lo:hi]).sum(0))], dtype=torch.float64)
    buf = pkg.dist.pack_partial(prec, pscal)
    buf = pkg.dist.all_reduce_partial(buf, pscal.shape[1])
    r, s = pkg.dist.unpack_partial(buf, prec.shape)
    fin = pkg.dist.finish_scalars(s)
    if rank == 0:
        q.put((r.numpy(), {k: v for k, v in fin.items()}, rec, nf, aik, flux, scal, tdg))
    dist.destroy_process_group()


def test_sharded_aggregate_matches_serial(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got_rec, fin, rec, nf, aik, flux, scal, tdg = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sb = np.zeros((len(aik), 7))
    sb[:, 0], sb[:, 1], sb[:, 2] = scal[:, 0], flux[:, 0], flux[:, 1]
    sb[:, 3:6] = scal[:, 1:4]
    exp_rec, exp_scal = oracle.aggregate(rec, nf.astype(np.int32), aik, sb)
    assert np.allclose(got_rec[0][:exp_rec.shape[0]], exp_rec, rtol=1e-12, atol=1e-14)
    assert np.allclose([fin["tdifmus"][0], fin["emoins"][0], fin["eplus"][0]], exp_scal[:3], rtol=1e-12)
    assert np.allclose([fin["ttot_tronc"][0], fin["ttot_vrai"][0], fin["tauout"][0]], exp_scal[3:6], rtol=1e-11)
    assert fin["n_orders"][0] == nf.max() and fin["min_orders"][0] == nf.min()
    assert np.allclose(fin["tdifmug"][0], (aik[:, None] * tdg).sum(0), rtol=1e-13)      # SOS_AGGREGATE.F:455-458
    assert abs(fin["sum_aik"][0] - 1.0) < 1e-12


def test_shard_range_partitions():
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    for nb in (1, 7, 32, 4097):
        for world in (1, 2, 3, 8):
            spans = [pkg.dist.shard_range(nb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks by itself (child torchrun) and report
    n_gpus = 2; --dry-run keeps it on the CPU (gloo, fabricated partials, the real pack / all-reduce / finish code)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--bins", "5"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["dry_run"] is True and res["reduce_ok"] is True


def test_bench_rejects_mismatched_world():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
