"""The sharded band solve on real GPUs, several processes (the multi-rank product path had only run with
fabricated partials on gloo).  torch.distributed.run starts the ranks; tests/dist_gpu_worker.py is one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, tmp_path, bins):
    out = str(tmp_path / "res.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--backend", backend, "--bins", str(bins),
           "--out", out]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return json.load(open(out))


@pytest.mark.gpu
@pytest.mark.parametrize("world,bins", [(2, 7), (3, 2)])
def test_sharded_band_on_one_gpu_several_processes(tmp_path, world, bins):
    """Ranks share cuda:0 and reduce through gloo: SosContext.solve_band per rank (uneven shards; with 3 ranks and 2 bins one
    rank holds NO bin and contributes the neutral element) equals the unsharded band to rounding of the summation order."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = _run(world, "gloo", tmp_path, bins)
    assert r["world"] == world and r["rec_err"] < 1e-13 and r["tdifmug_err"] < 1e-13 and r["flux_err"] < 1e-13
    assert r["n_orders"][0] == r["n_orders"][1] and abs(r["sum_aik"] - 1.0) < 1e-13 and r["min_orders"] > 0


@pytest.mark.gpu
def test_sharded_band_rccl_two_gpus(tmp_path):
    """One GPU per rank, RCCL all-reduce over xGMI (runs where two devices are visible)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _run(2, "nccl", tmp_path, 9)
    assert r["rec_err"] < 1e-13 and r["tdifmug_err"] < 1e-13 and r["n_orders"][0] == r["n_orders"][1]


@pytest.mark.gpu
def test_rccl_collectives_of_the_band_reduce_on_one_rank(tmp_path):
    """What the one-GPU box can tell about the RCCL path: a process group with backend "nccl" comes up, and the two collectives
    of dist.all_reduce_partial (fp64 SUM of the packed band, MAX of the order counts) run on device tensors of the band's size
    and leave a one-rank band unchanged.  (The two-rank test above needs two devices.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    code = (
        "import os, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "buf = torch.randn((1, 81 * 3 * 83 + 51), dtype=torch.float64, device='cuda', generator=g)\n"
        "ref = buf.clone(); mx = buf[:, -44:-42].clone()\n"
        "dist.all_reduce(buf, op=dist.ReduceOp.SUM); dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.barrier()\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(buf, ref) and torch.equal(mx, ref[:, -44:-42])\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "rccl ok" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("case,world", [("ckd_h2o_o2_25bins_flatsea", 3), ("ckd_o2a_5bins", 4), ("cfg2_lnd_lambert", 2)])
def test_sos_proc_sharded(tmp_path, case, world):
    """The drop-in under torch.distributed: run_sos.sos_proc shards the CKD bins of the band over the ranks (BASELINE config 3:
    25 bins over 3 ranks; 5 bins over 4 ranks with the Trans + Flux files written by rank 0 -- at most 4 ranks: the GPU box allows
    six processes on the card, this one included), joins them with
    the one all-reduce and returns the reference's outputs on every rank; a single-bin run (no gas absorption) is a replica on
    every rank and must not be reduced."""
    out = str(tmp_path / "res.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_proc_worker.py"), "--case", case, "--out", out]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.load(open(out))
    assert r["world"] == world and len(set(r["sums"])) == 1, r["sums"]
    assert "SOS_Result.bin" in r["files"] and "SOS_UsedAngles.txt" in r["files"]
