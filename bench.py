#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: CKD spectral bins/sec (full Stokes I,Q,U, TOA+surface).

Workload (BASELINE.json configs[1]): single wavelength, aerosol + Rayleigh atmosphere, 40 Gauss angles
(+ sun => N = 41 directions), 30 layers, OS_NB = 80 Fourier/Legendre orders, Lambertian surface, fp64,
synthetic seeded CKD bins (per-bin gas absorption).  One "step" = one pass of the hot path over
a batch of `--bins` bins per GPU already resident in HBM: fused SOS_OS solve of every bin + AIK-weighted
aggregation (+ one RCCL all-reduce of the band result when N > 1).  Bins are sharded over ranks with no
data-path collective besides that reduce; per-GPU work is fixed as N grows ("weak").

python bench.py --gpus N --steps K --warmup W     (N>1: launched by torch.distributed.run, one rank/GPU)
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X dense FP64 MFMA peak (MI355X_MICROARCH.md); scripts/ubench_mfma_peak.hip measures 78.1 on the box


def build_workload(S, nb, nt, seed, g):
    mu, w, n0 = S.gauss_angles(40, 35.0)
    os_nb = 80
    al, be, ga, ze = S.hg_phase(os_nb, g)
    bins = S.ckd_bins(nb, nt, seed=seed)
    h, x, y, iborm = S.rescale_profile(bins["h"], bins["xdel"], bins["ydel"], 0.0, 0.95, 0.95, os_nb)
    return dict(mu=mu, w=w, n0=n0, os_nb=os_nb, coefs=(al, be, ga, ze), h=h, xdel=x, ydel=y, zprof=bins["zprof"],
                aik=bins["aik"], iborm=iborm)


def cpu_baseline(wl, nsample):
    """Time the CPU reference on a bounded sample of the same bins (rank 0, N=1 only).  Uses the real
    reference Fortran (oracle/_ref/libsos_ref.so, built in the authoring container) when it loads,
    else the C restatement.  One core."""
    from oracle import oracle_ctypes, ref_ctypes
    kind, mod = "port", oracle_ctypes
    if ref_ctypes.available():
        try:
            ref_ctypes.lib()
            kind, mod = "reference", ref_ctypes
        except OSError:
            pass
    al, be, ga, ze = wl["coefs"]
    kw = dict(n0=wl["n0"], ro=0.1, iborm=wl["iborm"])
    if kind == "reference":
        kw["want_log"] = False
    t0 = time.perf_counter()
    done = 0
    for b in range(nsample):
        mod.sos_os(wl["mu"], wl["w"], wl["os_nb"], wl["h"][b], wl["xdel"][b], wl["ydel"][b], al, be, ga, ze,
                   zprof=wl["zprof"][b], **kw)
        done += 1
        if time.perf_counter() - t0 > 40.0:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="bins/s", cores=1, kind=kind,
                sample="first %d bins of the bench batch, serial SOS_OS calls (%s), %.1f s" % (
                    done, "amdflang -O2 build of the reference Fortran" if kind == "reference" else "C restatement -O2", dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=4096, help="CKD bins per GPU per step")
    ap.add_argument("--nt", type=int, default=30)
    ap.add_argument("--g", type=float, default=0.75)
    ap.add_argument("--cpu-sample", type=int, default=224, help="bins timed on the CPU baseline (about 15 s of one core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-sort", action="store_true", help="keep the bins in generation order (default: cost-sorted upload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    S = pkg.synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = args.gpus > 1 or world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()

    # every rank owns `bins` bins of one global band of world*bins bins (weights normalised globally)
    nb_tot = args.bins * world
    wl = build_workload(S, nb_tot, args.nt, 1234, args.g)
    lo, hi = pkg.dist.shard_range(nb_tot, rank, world)
    al, be, ga, ze = wl["coefs"]
    cx = pkg.SosContext(wl["mu"], wl["w"], wl["n0"], al, be, ga, ze, iborm_max=wl["iborm"], ro=0.1, device=dev)
    bins = cx.upload_bins(wl["h"][lo:hi], wl["xdel"][lo:hi], wl["ydel"][lo:hi], order=None if args.no_sort else "cost")
    aik_h = wl["aik"][lo:hi].copy()
    if bins["perm"] is not None:
        aik_h = aik_h[bins["perm"]]
    aik = torch.from_numpy(aik_h).to(cx.device)
    out = cx.alloc_outputs(hi - lo)
    torch.cuda.synchronize()

    def step():
        cx.solve(bins, out)
        rec, scal = cx.aggregate(out, aik)
        buf = pkg.dist.pack_partial(rec, scal)
        return pkg.dist.all_reduce_partial(buf)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kms.append(None)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel (k_sos_os): HIP-event duration of the LAST timed launch on its own
    # stream + a few extra launches for an average, algorithmic flops counted from the run's own
    # Fourier/scattering-order counts (SURVEY 8d W_step).
    durs = []
    for _ in range(3):
        cx.solve(bins, out)
        durs.append(cx.last_solve_ms())
    kern_ms = float(np.mean(durs))
    flops_ref, flops_exe = cx.solve_flops(bins, out)
    achieved = flops_exe / (kern_ms * 1e-3) / 1e12
    # HBM traffic of one launch: rocprofv3 PMC passes of this same command, summarised (with the gfx950 corrections
    # of MI355X_MICROARCH.md) by scripts/summarize_profiles.py into profiles/; null when the workload differs
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")) as f:
            pm = json.load(f)
        if pm.get("bins_per_gpu") == args.bins and pm.get("nt") == args.nt:
            traffic = pm["k_sos_os_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    res = dict(metric="CKD spectral bins/sec (full Stokes I,Q,U, TOA+surface)", value=nb_tot * args.steps / dt,
               unit="bins/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * dt / args.steps,
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="single-wavelength aerosol+Rayleigh, 40 Gauss angles (N=41), NT=%d layers, OS_NB=80, "
                                    "Lambertian rho=0.1, HG g=%.2f, %d CKD bins/GPU/step" % (args.nt, args.g, args.bins),
                           bins_per_gpu=args.bins, nt=args.nt, n_dirs=41, os_nb=80, parallelism="bins sharded x%d" % world,
                           bin_order="generation" if args.no_sort else "cost-sorted"),
               roofline=dict(bound="mfma", achieved=achieved, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=achieved / FP64_PEAK_TFLOPS, traffic=traffic,
                             kernel="k_sos_os<4,2,2,false,false,false>", kernel_ms=kern_ms, flops_per_launch=flops_exe,
                             flops_counted="parity form (two 3N x 3Nw half systems, Nw = weighted directions) + rank-4 molecular form + formal solution, unpadded",
                             reference_algorithm_flops_per_launch=flops_ref,
                             reference_algorithm_tflops=flops_ref / (kern_ms * 1e-3) / 1e12))
    if rank == 0:
        nord = out["norders"].cpu().numpy()
        res["config"]["mean_fourier_orders"] = float(nord.mean())
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample)
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
