#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: CKD spectral bins/sec (full Stokes I,Q,U, TOA+surface).

Workload (BASELINE.json configs[1]): single wavelength, aerosol + Rayleigh atmosphere, 40 Gauss angles
(+ sun => N = 41 directions), 30 layers, OS_NB = 80 Fourier/Legendre orders, Lambertian surface, fp64,
synthetic seeded CKD bins (per-bin gas absorption).  One "step" = one pass of the hot path over
a batch of `--bins` bins per GPU already resident in HBM: fused SOS_OS solve of every bin + AIK-weighted
aggregation (+ one RCCL all-reduce of the band result when N > 1).  Bins are sharded over ranks with no
data-path collective besides that reduce; per-GPU work is fixed as N grows ("weak").

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (WORLD_SIZE unset) the script starts its own N ranks through
`python -m torch.distributed.run` as a CHILD process, before anything touches the GPU, and exits with the child's
code; under a launcher (the driver's torch.distributed.run) it is one rank.  Rank 0 prints ONE JSON line.

--workload realistic (also appended to the default N = 1 line as "realistic_mix"): the same wavelength with the level
grids SOS_PROFILE really produces (NT 117...426 from seeded gas columns, made on the device by sosgpu_profile), i.e. the
streamed-field variant of the solver.
--workload hyperspectral (also appended to the default N = 1 line as "hyperspectral"): BASELINE config 5 through the drop-in --
run_sos.sos_spectrum over the 2496 spectral intervals of 2500-27500 cm-1 (synthetic CKD tables with the reference's own
number of exponential terms per gas and interval, scripts/synth_ckd.py; 5732 bins), wavelengths dealt to the ranks.
--scaling strong: ONE global band of --bins bins (headline) split over the N ranks by cost (dist.balanced_shards), the
per-rank solve / reduce times in the line, so that imbalance is visible; default "weak": --bins bins per rank.
--dry-run: CPU rehearsal of the multi-rank path (gloo, fabricated partials, no solver) used by tests/test_dist_cpu.py.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SMALL_BATCH = 4096          # round 1's bins per step (kept as a second measurement; the realistic mix uses it too)
FP64_PEAK_TFLOPS = 78.6   # MI355X dense FP64 MFMA peak (MI355X_MICROARCH.md); scripts/ubench_mfma_peak.hip measures 78.1 on the box
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md


def build_workload(S, nb, nt, seed, g, part=None):
    """The seeded band of nb bins; part = (lo, hi): only this rank's slice of it is built."""
    mu, w, n0 = S.gauss_angles(40, 35.0)
    os_nb = 80
    al, be, ga, ze = S.hg_phase(os_nb, g)
    bins = S.ckd_bins(nb, nt, seed=seed, part=part)
    h, x, y, iborm = S.rescale_profile(bins["h"], bins["xdel"], bins["ydel"], 0.0, 0.95, 0.95, os_nb)
    return dict(mu=mu, w=w, n0=n0, os_nb=os_nb, coefs=(al, be, ga, ze), h=h, xdel=x, ydel=y, zprof=bins["zprof"],
                aik=bins["aik"], iborm=iborm)


def realistic_columns(nb, seed=5):
    """Seeded gas columns of the realistic mix (k_b log-uniform in [1e-3, 30]) on 50 descending altitudes:
    cumulative absorption optical depth per bin, as SOS_ABSPROFILE hands it to SOS_PROFILE."""
    alt = np.concatenate([np.linspace(120.0, 30.0, 10), np.linspace(28.0, 0.0, 40)])
    col = np.exp(-alt / 7.0)
    col[0] = 0.0
    rng = np.random.default_rng(seed)
    scale = np.exp(rng.uniform(np.log(1e-3), np.log(30.0), nb))
    return alt, scale[:, None] * col[None, :]


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


def cpu_worker(args):
    """One process of the all-cores CPU baseline: the same serial SOS_OS calls as cpu_baseline on a slice of the bench batch,
    for about `--cpu-seconds` seconds; prints `done elapsed kind`."""
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    from oracle import oracle_ctypes, ref_ctypes
    kind, mod = "port", oracle_ctypes
    if ref_ctypes.available():
        try:
            ref_ctypes.lib()
            kind, mod = "reference", ref_ctypes
        except OSError:
            pass
    nb = 96
    wl = build_workload(pkg.synth, args.bins, args.nt, 1234, args.g, part=(args.cpu_worker * nb, (args.cpu_worker + 1) * nb))
    al, be, ga, ze = wl["coefs"]
    kw = dict(n0=wl["n0"], ro=0.1, iborm=wl["iborm"])
    if kind == "reference":
        kw["want_log"] = False
    mod.sos_os(wl["mu"], wl["w"], wl["os_nb"], wl["h"][0], wl["xdel"][0], wl["ydel"][0], al, be, ga, ze, zprof=wl["zprof"][0], **kw)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        b = done % nb
        mod.sos_os(wl["mu"], wl["w"], wl["os_nb"], wl["h"][b], wl["xdel"][b], wl["ydel"][b], al, be, ga, ze, zprof=wl["zprof"][b], **kw)
        done += 1
    print("CPUWORKER %d %.3f %s" % (done, time.perf_counter() - t0, kind))


def cpu_baseline_all_cores(seconds):
    """The reference on ALL host cores of the box: one PROCESS per core, each with its own temporary directory (the Fortran is
    not re-entrant: fixed unit numbers and temporary file names, SOS_PROC.F:1460-1466 -- so processes, not threads), the
    serial SOS_OS calls of cpu_baseline on different slices of the bench batch, all running for the same wall time.
    Started before this process touches the GPU; collected by finish_cpu_all_cores()."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # the GPU box gives one GPU's job a share of 16 host cores (its affinity mask shows the whole machine): stay inside it
    cores = max(1, min(cores, int(os.environ.get("SOS_BENCH_CPU_CORES", "16"))))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--cpu-seconds", str(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(cores)]
    return procs


def finish_cpu_all_cores(procs):
    done, tmax, kind = 0, 0.0, "port"
    for p in procs:
        out, _ = p.communicate(timeout=300)
        for ln in out.splitlines():
            if ln.startswith("CPUWORKER"):
                _, d, t, kind = ln.split()
                done += int(d)
                tmax = max(tmax, float(t))
    if not tmax:
        return None
    return dict(value=done / tmax, unit="bins/s", cores=len(procs), kind=kind,
                sample="%d processes (one per host core), each serial SOS_OS calls on its own 96-bin slice of the bench batch for "
                       "%.1f s: %d bins in all" % (len(procs), tmax, done))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a child torchrun and relay its exit
    code.  Nothing in this process has touched the GPU yet (no torch import)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % (args.gpus * args.ranks_per_gpu),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """CPU rehearsal of the N-rank path: process group (gloo), shard ranges, packed partials, the one SUM all-reduce +
    the MAX pair, finish.  No solver: partials are fabricated from a seeded generator, so rank 0 can check the reduce."""
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, s1 = 5, 4
    w = 2 * n + 1
    nb_tot = args.bins * world
    rng = np.random.default_rng(7)
    rec_all = rng.normal(size=(nb_tot, s1, 3, w))
    aik = rng.dirichlet(np.ones(nb_tot))
    nord = rng.integers(1, s1 + 1, nb_tot)
    lo, hi = pkg.dist.shard_range(nb_tot, rank, world)
    prec = torch.from_numpy((aik[lo:hi, None, None, None] * rec_all[lo:hi]).sum(0, keepdims=True))
    sc = np.zeros((1, 10 + n))
    sc[0, 6], sc[0, 7], sc[0, 8] = aik[lo:hi].sum(), (nord[lo:hi].max() if hi > lo else 0), (-nord[lo:hi].min() if hi > lo else -2147483647.)
    sc[0, 3:6] = aik[lo:hi].sum()
    buf = pkg.dist.all_reduce_partial(pkg.dist.pack_partial(prec, torch.from_numpy(sc)), 10 + n)
    rec, scal = pkg.dist.unpack_partial(buf, prec.shape)
    fin = pkg.dist.finish_scalars(scal)
    ok = bool(np.allclose(rec[0].numpy(), (aik[:, None, None, None] * rec_all).sum(0), rtol=1e-12, atol=1e-14)
              and abs(fin["sum_aik"][0] - 1.0) < 1e-12 and fin["n_orders"][0] == nord.max() and fin["min_orders"][0] == nord.min())
    if rank == 0:
        print(json.dumps(dict(metric="CKD spectral bins/sec (full Stokes I,Q,U, TOA+surface)", value=None, unit="bins/s",
                              n_gpus=world, steps=0, warmup=0, dry_run=True, reduce_ok=ok,
                              config=dict(workload="dry run (gloo, fabricated partials)", bins_per_gpu=args.bins))))
    dist.destroy_process_group()
    return 0 if ok else 1


def timed_steps(step, fence, steps, warmup, world, dist, device, torch):
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_ms(cx, bins, out, reps=3):
    durs = []
    for _ in range(reps):
        cx.solve(bins, out)
        durs.append(cx.last_solve_ms())
    return float(np.mean(durs))


def pmc_traffic(tag, **match):
    """HBM traffic of one launch: rocprofv3 PMC passes of this same command, summarised (with the gfx950 corrections of
    MI355X_MICROARCH.md) by scripts/summarize_profiles.py into profiles/; None when the workload differs."""
    for rnd in ("r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_pmc_hbm%s.json" % (rnd, tag))) as f:
                pm = json.load(f)
            if all(pm.get(k) == v for k, v in match.items()):
                return pm["k_sos_os_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            continue
    return None


def run_realistic(pkg, torch, dist, world, rank, nbins, steps, warmup, g, strong=False):
    """The level grids real CKD bins get (NT >= CTE_OS_NT_MIN = 100, SOS.h:229): profiles made on the device by
    sosgpu_profile from seeded gas columns, solved by the streamed-field variant.  Returns the result dict."""
    S = pkg.synth
    mu, w, n0 = S.gauss_angles(40, 35.0)
    os_nb = 80
    al, be, ga, ze = S.hg_phase(os_nb, g)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=os_nb, ro=0.1)
    strong = strong and world > 1
    nb_tot = nbins if strong else nbins * world
    alt, tabs = realistic_columns(nb_tot)
    aik_all = np.random.default_rng(1234).dirichlet(np.ones(nb_tot))
    if strong:          # one band dealt to the ranks by cost (levels x scattering steps, dist.bin_cost)
        sel = pkg.dist.balanced_shards(pkg.dist.bin_cost(0.0948 + 0.3, tabs[:, -1]), world)[rank]
    else:
        sel = np.arange(*pkg.dist.shard_range(nb_tot, rank, world))
    order = np.argsort(-tabs[sel, -1], kind="stable")             # cost-sorted (thickest gas column first)
    bins = cx.make_profiles(len(sel), 0.0948, 8.0, 0.3, 2.0, alt, tabs[sel][order], piz=0.95, piztr=0.95)
    aik = torch.from_numpy(aik_all[sel][order]).to(cx.device)
    out = cx.alloc_outputs(len(sel))
    torch.cuda.synchronize()

    def step():
        cx.solve(bins, out)
        rec, scal = cx.aggregate(out, aik, scal=bins["scal"])
        return pkg.dist.all_reduce_partial(pkg.dist.pack_partial(rec, scal), scal.shape[1])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt = timed_steps(step, fence, steps, warmup, world, dist, cx.device, torch)
    kms = kernel_ms(cx, bins, out, 2)
    flops_ref, flops_exe = cx.solve_flops(bins, out)
    nt = bins["nt"].cpu().numpy().astype(np.int64)
    igl = out["iglast"].cpu().numpy().astype(np.int64)
    nsteps = np.clip(igl - 1, 0, None).sum(axis=1)                # contraction steps (scattering orders >= 2) per bin
    passes = np.clip(igl, 0, None).sum(axis=1)                    # field passes per bin: every scattering order, order 1 included
    # algorithmic field traffic of the streamed variant: each pass writes the 6N-row field of NT+1 levels once and every
    # pass but the first of a Fourier order reads it once
    field = (nt + 1) * 6 * cx.n * 8.0
    nord = out["norders"].cpu().numpy().astype(np.int64)
    bytes_alg = float((field * (2 * passes - np.clip(nord, 0, None))).sum())
    ach = bytes_alg / (kms * 1e-3) / 1e9
    tf = flops_exe / (kms * 1e-3) / 1e12
    res = dict(value=nb_tot * steps / dt, unit="bins/s", ms_per_step=1e3 * dt / steps, steps=steps, warmup=warmup,
               config=dict(workload="same wavelength as the headline, level grids of SOS_PROFILE for seeded gas columns "
                                    "(k log-uniform 1e-3..30): NT %d...%d, mean %.0f; %d bins/GPU/step" % (
                                        nt.min(), nt.max(), nt.mean(), nbins),
                           bins_per_gpu=len(sel), nt_min=int(nt.min()), nt_mean=float(nt.mean()), nt_max=int(nt.max()),
                           mean_fourier_orders=float(nord.mean()), mean_scattering_steps=float(nsteps.mean())),
               roofline=dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                             traffic=pmc_traffic("_realistic", bins_per_gpu=nbins),
                             kernel="k_sos_stream<4,2,false,false,false>", kernel_ms=kms, bytes_per_launch=bytes_alg,
                             bytes_counted="field written once per scattering order and read once per order >= 2: "
                                           "(NT+1) x 6N x 8 B each",
                             mfma_tflops=tf, mfma_frac=tf / FP64_PEAK_TFLOPS))
    cx.close()
    return res


def small_band_latency(pkg, torch, g, nbins=25):
    """One wavelength's band as a per-call user has it: `nbins` CKD bins on real level grids, ONE solve + aggregate, wall clock
    (the order-parallel form of the streamed solver: the Fourier orders of a bin as independent workgroups)."""
    S = pkg.synth
    mu, w, n0 = S.gauss_angles(40, 35.0)
    al, be, ga, ze = S.hg_phase(80, g)
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=80, ro=0.1)
    alt, tabs = realistic_columns(nbins)
    bins = cx.make_profiles(nbins, 0.0948, 8.0, 0.3, 2.0, alt, tabs, piz=0.95, piztr=0.95)
    aik = torch.full((nbins,), 1.0 / nbins, dtype=torch.float64, device=cx.device)
    out = cx.alloc_outputs(nbins)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cx.solve(bins, out)
        cx.aggregate(out, aik, scal=bins["scal"])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    nt = bins["nt"].cpu().numpy()
    cx.close()
    return dict(bins=nbins, ms_per_solve=1e3 * min(ts[1:]), value=nbins / min(ts[1:]), unit="bins/s",
                config="one band of %d bins, level grids NT %d...%d, N = 41, OS_NB = 80: solve + aggregate, wall clock, min of 5"
                       % (nbins, nt.min(), nt.max()))


def measure_headline(pkg, S, torch, dist, world, rank, dev, nbins, steps, warmup, args, base):
    """One measurement of the headline workload: returns (result dict, workload, per-bin Fourier-order counts).
    weak scaling: every rank owns `nbins` bins of one global band of world * nbins bins (weights normalised globally);
    strong scaling (args.scaling): ONE band of `nbins` bins, dealt to the ranks by cost (dist.balanced_shards)."""
    strong = getattr(args, "scaling", "weak") == "strong" and world > 1
    if strong:
        nb_tot = nbins
        full = S.ckd_bins(nb_tot, args.nt, seed=1234)
        costs = pkg.dist.bin_cost(0.3948, np.maximum(full["h"][:, -1] - 0.3948, 0.0), fixed_levels=True)
        mine = pkg.dist.balanced_shards(costs, world)[rank]
        wl = build_workload(S, nb_tot, args.nt, 1234, args.g)
        for k in ("h", "xdel", "ydel", "zprof", "aik"):
            wl[k] = wl[k][mine]
        nloc = len(mine)
    else:
        nb_tot = nbins * world
        lo, hi = pkg.dist.shard_range(nb_tot, rank, world)
        wl = build_workload(S, nb_tot, args.nt, 1234, args.g, part=(lo, hi))
        nloc = hi - lo
    al, be, ga, ze = wl["coefs"]
    cx = pkg.SosContext(wl["mu"], wl["w"], wl["n0"], al, be, ga, ze, iborm_max=wl["iborm"], ro=0.1, device=dev)
    bins = cx.upload_bins(wl["h"], wl["xdel"], wl["ydel"], order=None if args.no_sort else "cost")
    aik_h = wl["aik"].copy()
    if bins["perm"] is not None:
        aik_h = aik_h[bins["perm"]]
    aik = torch.from_numpy(aik_h).to(cx.device)
    out = cx.alloc_outputs(nloc)
    torch.cuda.synchronize()

    def step():
        cx.solve(bins, out)
        rec, scal = cx.aggregate(out, aik)
        buf = pkg.dist.pack_partial(rec, scal)
        return pkg.dist.all_reduce_partial(buf, scal.shape[1])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt = timed_steps(step, fence, steps, warmup, world, dist, cx.device, torch)

    # roofline of the dominant kernel (k_sos_os): HIP-event duration on its own stream (sosgpu_last_solve_ms), averaged
    # over a few launches; algorithmic flops counted from the run's own Fourier/scattering-order counts.
    kern_ms = kernel_ms(cx, bins, out)
    flops_ref, flops_exe = cx.solve_flops(bins, out)
    achieved = flops_exe / (kern_ms * 1e-3) / 1e12
    traffic = pmc_traffic("", bins_per_gpu=nbins, nt=args.nt)

    res = dict(base, value=nb_tot * steps / dt, steps=steps, warmup=warmup, ms_per_step=1e3 * dt / steps,
               config=dict(workload="single-wavelength aerosol+Rayleigh, 40 Gauss angles (N=41), NT=%d layers, OS_NB=80, "
                                    "Lambertian rho=0.1, HG g=%.2f, %d CKD bins/GPU/step" % (args.nt, args.g, nbins),
                           bins_per_gpu=nloc if strong else nbins, nt=args.nt, n_dirs=41, os_nb=80, parallelism="bins sharded x%d" % world,
                           bin_order="generation" if args.no_sort else "cost-sorted"),
               roofline=dict(bound="mfma", achieved=achieved, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=achieved / FP64_PEAK_TFLOPS, traffic=traffic,
                             kernel="k_sos_os<4,2,2,false,false>", kernel_ms=kern_ms, flops_per_launch=flops_exe,
                             flops_counted="parity form (two 3N x 3Nw half systems, Nw = weighted directions) + rank-4 molecular form + formal solution, unpadded",
                             reference_algorithm_flops_per_launch=flops_ref,
                             reference_algorithm_tflops=flops_ref / (kern_ms * 1e-3) / 1e12))
    if world > 1:
        # per-rank solve and reduce times (imbalance shows here): kernel time of the rank's own solve from its HIP events, the
        # band all-reduce timed on its own after a barrier
        rec, scal = cx.aggregate(out, aik)
        buf = pkg.dist.pack_partial(rec, scal)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            pkg.dist.all_reduce_partial(buf, scal.shape[1])
        e1.record()
        torch.cuda.synchronize()
        mine_t = torch.tensor([kern_ms, e0.elapsed_time(e1) / 5.0, float(nloc)], dtype=torch.float64, device=cx.device)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        res["per_rank"] = dict(solve_ms=[float(t[0]) for t in allt], reduce_ms=[float(t[1]) for t in allt],
                               bins=[int(t[2]) for t in allt])
    res["scaling"] = "strong" if strong else "weak"
    if strong:
        res["config"]["parallelism"] = "one band of %d bins dealt to %d ranks by cost" % (nb_tot, world)
    nord = out["norders"].cpu().numpy()
    cx.close()
    return res, wl, nord


def measure_variant(pkg, S, torch, name, nbins, kw, nt=30, g=0.75, ng=40):
    """The headline workload with another surface (BASELINE configs 3 and 4: the SURF / Fresnel variants of k_sos_os), another
    number of Gauss angles or levels (the reference's default of 24 angles: other kernel instantiations), kernel time from HIP
    events, executed-flops roofline fraction."""
    mu, w, n0 = S.gauss_angles(ng, 35.0)
    os_nb = 80
    al, be, ga, ze = S.hg_phase(os_nb, g)
    b = S.ckd_bins(nbins, nt, seed=1234)
    h, x, y, iborm = S.rescale_profile(b["h"], b["xdel"], b["ydel"], 0.0, 0.95, 0.95, os_nb)
    if kw.get("imat_surf"):
        kw = dict(kw, rsurf=pkg.surface.glitter_matrices(mu, w, 7.0, 1.34, iborm, os_nb, 2 * os_nb)["rsurf"][:iborm + 1])
    cx = pkg.SosContext(mu, w, n0, al, be, ga, ze, iborm_max=iborm, **kw)
    bins = cx.upload_bins(h, x, y, order="cost")
    out = cx.alloc_outputs(nbins)
    cx.solve(bins, out)
    torch.cuda.synchronize()
    kms = kernel_ms(cx, bins, out)
    _, flops_exe = cx.solve_flops(bins, out)
    tf = flops_exe / (kms * 1e-3) / 1e12
    res = dict(workload=name, value=nbins / (kms * 1e-3), unit="bins/s", bins=nbins, kernel_ms=kms, mfma_tflops=tf,
               roofline_frac=tf / FP64_PEAK_TFLOPS, mean_fourier_orders=float(out["norders"].float().mean()))
    cx.close()
    return res


def hyperspectral_kwargs(rs, every=1):
    """sos_proc keyword sets of BASELINE config 5: one call per 10 cm-1 interval of 2500-27500 cm-1 (the wavelengths SOS_PROC
    accepts, 0.364-4 um), all gases, mid-latitude summer + user CO2 / CH4, log-normal aerosol, Roujean + Maignan surface."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import hyperspectral_bench
    return hyperspectral_bench.spectrum_kwargs(rs, every)


def run_hyperspectral(pkg, torch, dist, world, rank, every=1):
    """BASELINE config 5 through run_sos.sos_spectrum; with N ranks the wavelengths are dealt to the ranks by cost and the
    results gathered (no all-reduce).  Synthetic CKD tables are written once per rank (scripts/synth_ckd.py)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import synth_ckd
    rs = pkg.run_sos
    root = tempfile.mkdtemp(prefix="synth_fic_")
    synth_ckd.write_tables(root)
    os.environ["SOS_ABS_ROOT"] = root
    kws = hyperspectral_kwargs(rs, every)
    nb = sum(pkg.absorption.band_bin_count(kw["wa_simu"], 10.0) for kw in kws)
    rs.sos_spectrum(kws[:16], gather=False)                  # warm-up: library, surface matrices, Mie records, table parsing
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    rs.sos_spectrum(kws, timings=tm, gather=world > 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nmine = max(1, len(kws) // world)
    host = (tm["prepare"] + tm["solve_launch"] + tm["finish"]) / nmine
    wait = (tm["wait"] + tm["trphi"]) / nmine
    return dict(value=nb / dt, unit="bins/s", wavelengths_per_s=len(kws) / dt, wavelengths=len(kws), bins=nb, seconds=dt,
                host_ms_per_wavelength=1e3 * host, gpu_wait_ms_per_wavelength=1e3 * wait,
                host_phases_ms_per_wavelength={k: 1e3 * v / nmine for k, v in tm.items()},
                config=dict(workload="hyperspectral 2500-27500 cm-1 in 10 cm-1 intervals through run_sos.sos_spectrum: %d "
                                     "wavelengths, %d CKD bins (1...125 per wavelength; synthetic tables with the reference's "
                                     "exponential-term counts), all gases, LND aerosol by Mie theory per wavelength, Roujean + "
                                     "Maignan surface, 16 Gauss angles, polar view" % (len(kws), nb),
                            parallelism="wavelengths dealt to %d rank(s) by cost, results gathered" % world))


def hyperspectral_four_processes(args, user_queues):
    """The hyperspectral workload with FOUR host processes sharing the GPU (`--workload hyperspectral --ranks-per-gpu 4` as a
    child job: the wavelength partition of the multi-GPU path, gloo gather): the preparation of a wavelength is ~1.2 ms of
    Python against ~0.1 ms of GPU time, so one interpreter per GPU leaves the card idle.  Returns the child's figures or None."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "hyperspectral", "--ranks-per-gpu", "4",
           "--spectrum-every", str(args.spectrum_every)]
    env = dict(os.environ)
    if user_queues is None:
        env.pop("GPU_MAX_HW_QUEUES", None)                  # (the child picks the queue count of its own layout)
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, env=env).stdout.decode()
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {k: r[k] for k in ("value", "unit", "wavelengths_per_s", "seconds", "host_ms_per_wavelength",
                                  "gpu_wait_ms_per_wavelength", "host_processes_per_gpu")}
    except Exception as e:                                  # (the one-process figure stands on its own)
        return dict(error=str(e)[:200])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=32768, help="CKD bins per GPU per step (records: 161 KB of HBM per bin)")
    ap.add_argument("--nt", type=int, default=30)
    ap.add_argument("--g", type=float, default=0.75)
    ap.add_argument("--workload", choices=["headline", "realistic", "hyperspectral"], default="headline")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --bins bins per GPU; strong: one band of --bins bins split over the GPUs by cost")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall time of the all-cores CPU baseline")
    ap.add_argument("--no-hyper", action="store_true", help="skip the appended hyperspectral measurement (N = 1 default run)")
    ap.add_argument("--spectrum-every", type=int, default=1, help="hyperspectral: take every k-th spectral interval")
    ap.add_argument("--ranks-per-gpu", type=int, default=1,
                    help="hyperspectral only: host processes (ranks) sharing each GPU -- the wavelength partition is the same as "
                         "over GPUs, the host preparation (Python, one interpreter per rank) runs in parallel; gloo gather")
    ap.add_argument("--cpu-sample", type=int, default=224, help="bins timed on the CPU baseline (about 15 s of one core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mix", action="store_true", help="skip the appended realistic_mix measurement (N = 1 default run)")
    ap.add_argument("--no-sort", action="store_true", help="keep the bins in generation order (default: cost-sorted upload)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the multi-rank path (gloo, no solver)")
    args = ap.parse_args()

    if args.cpu_worker >= 0:
        cpu_worker(args)
        return
    # sos_spectrum spreads the per-wavelength preparation kernels over HIP streams: give them hardware queues of their own (the
    # runtime's default is 4; must be set before the first GPU call of the process)
    rpg = max(1, args.ranks_per_gpu)
    user_queues = os.environ.get("GPU_MAX_HW_QUEUES")
    # (several host processes on one card share about sixteen hardware queues: beyond that the card's queues are oversubscribed
    #  and every launch waits for a queue switch -- four processes: 1598 / 2306 / 2024 / 1363 wavelengths/s with 2 / 4 / 6 / 8
    #  queues each, profiles/r03_hyperspectral.txt)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(2, 16 // rpg)))
    if rpg > 1 and args.workload != "hyperspectral":
        sys.exit("bench.py: --ranks-per-gpu is for --workload hyperspectral (the bin-sharded workloads fill a GPU from one rank)")
    if args.gpus * rpg > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) // rpg
    if world != args.gpus * rpg:
        sys.exit("bench.py: --gpus %d x --ranks-per-gpu %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, rpg, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))
    # the all-cores CPU baseline runs in child processes started BEFORE this process initialises the GPU, while the main
    # process imports torch and builds its workload (its ~12 s overlap the GPU warm-up, not the timed steps: collected first)
    cpu_procs = None
    hyper_procs4 = None
    if world == 1 and rank == 0 and args.workload == "headline" and not args.no_hyper and not args.no_mix:
        import torch                                   # (pages the library in for the child ranks too; no GPU call yet)
        hyper_procs4 = hyperspectral_four_processes(args, user_queues)         # (a child job, finished before this process uses the GPU)
    if world == 1 and rank == 0 and not args.no_cpu and args.workload == "headline":
        cpu_procs = cpu_baseline_all_cores(args.cpu_seconds)

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    S = pkg.synth
    if world > 1:
        torch.cuda.set_device(local)
        # several ranks on one GPU: RCCL refuses duplicate devices, and the wavelength partition only gathers host objects
        dist.init_process_group("gloo" if rpg > 1 else "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    base = dict(metric="CKD spectral bins/sec (full Stokes I,Q,U, TOA+surface)", unit="bins/s", n_gpus=args.gpus,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic")

    if args.workload == "realistic":
        r = run_realistic(pkg, torch, dist, world, rank, args.bins, args.steps, args.warmup, args.g, strong=args.scaling == "strong")
        r["config"]["parallelism"] = "bins sharded x%d" % world
        if args.scaling == "strong" and world > 1:
            r["scaling"] = "strong"
        if rank == 0:
            print(json.dumps(dict(base, **r)))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "hyperspectral":
        r = run_hyperspectral(pkg, torch, dist, world, rank, args.spectrum_every)
        if rpg > 1:
            r["host_processes_per_gpu"] = rpg
        if rank == 0:
            print(json.dumps(dict(base, scaling="strong", steps=1, warmup=1, ms_per_step=1e3 * r["seconds"], **r)))
        if world > 1:
            dist.destroy_process_group()
        return
    cpu_all = finish_cpu_all_cores(cpu_procs) if cpu_procs else None      # (before the timed steps: the host is quiet again)

    res, wl, nord = measure_headline(pkg, S, torch, dist, world, rank, dev, args.bins, args.steps, args.warmup, args, base)
    if world == 1 and not args.no_mix:
        if args.bins != SMALL_BATCH:
            # the same workload in round 1's batch size: one dispatch round of the chip is 512 bins, so a 4096-bin step spends
            # about 5 % of its time in the tail of its last round; kept for continuity with earlier rounds
            sm, _, _ = measure_headline(pkg, S, torch, dist, world, rank, dev, SMALL_BATCH, max(5, args.steps), 2, args, {})
            res["batch_4096"] = dict(value=sm["value"], unit="bins/s", ms_per_step=sm["ms_per_step"], steps=sm["steps"],
                                     bins_per_gpu=SMALL_BATCH, roofline_frac=sm["roofline"]["frac"],
                                     kernel_ms=sm["roofline"]["kernel_ms"])
        res["realistic_mix"] = run_realistic(pkg, torch, dist, world, rank, SMALL_BATCH, max(3, args.steps // 3), 2, args.g)
        res["small_band"] = small_band_latency(pkg, torch, args.g)
        # BASELINE configs 3 and 4 on the same synthetic band: the Fresnel and the surface-matrix (SURF) variants of k_sos_os
        res["variants"] = [measure_variant(pkg, S, torch, "cfg3 flat sea (Fresnel interface, n = 1.34)", SMALL_BATCH,
                                           dict(ro=0.02, ifresnel=1, ind_surf=1.34)),
                           measure_variant(pkg, S, torch, "cfg4 Cox-Munk glitter 7 m/s (surface matrices, ground_mfma)", SMALL_BATCH,
                                           dict(ro=0.0, imat_surf=1)),
                           # the reference's default angle count (24 Gauss angles, N = 25): the shared-tile LDS variant and the
                           # five-tile layout of the streamed kernel
                           measure_variant(pkg, S, torch, "N = 25 (24 Gauss angles), NT = 30: k_sos_os<4,2,2,split>", SMALL_BATCH,
                                           dict(ro=0.1), ng=24),
                           measure_variant(pkg, S, torch, "N = 25 (24 Gauss angles), NT = 120: k_sos_stream<4,2,KHT=5>", SMALL_BATCH,
                                           dict(ro=0.1), nt=120, ng=24)]
        if not args.no_hyper:
            res["hyperspectral"] = run_hyperspectral(pkg, torch, dist, world, rank, args.spectrum_every)
            if hyper_procs4:
                res["hyperspectral"]["four_host_processes"] = hyper_procs4
    if rank == 0:
        res["config"]["mean_fourier_orders"] = float(nord.mean())
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample)
            if cpu_all:
                res["cpu_baseline"]["all_cores"] = cpu_all
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
