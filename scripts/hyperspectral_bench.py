"""BASELINE config 5 through the drop-in: a hyperspectral run, 2500-27500 cm-1 in 10 cm-1 intervals (the wavelengths SOS_PROC
accepts: 0.364-4 um, 2497 calls), all eight gases with the reference's own number of CKD terms per gas and interval (synthetic
coefficient values, scripts/synth_ckd.py: 5733 bins, 1...125 per wavelength), mid-latitude summer + user CO2 / CH4, log-normal
aerosol (Mie theory per wavelength), Roujean BRDF + Maignan BPDF, 16 Gauss angles, polar view.

Measures, on one GPU: (a) the plain loop of run_sos.sos_proc calls (what the reference's front end does, binding/run_sos.py:640),
(b) run_sos.sos_proc_many (host threads + streams), (c) run_sos.sos_spectrum (one launch per kernel variant), with the host
phases of (c) and, with --profile, a cProfile of it.  --n limits the spectrum to every k-th interval."""
import argparse, cProfile, importlib, io, os, pstats, sys, tempfile, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # hardware queues for the side streams of sos_spectrum (runtime default: 4)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import synth_ckd

USER = {"-ANG.Thetas": 35.0, "-AP.HR": 8.0, "-AP.AerHS.HA": 2.0, "-AP.SpectralResol": 10.0, "-AP.Psurf": 1013.0, "-AER.Waref": 0.55,
        "-SOS.IGmax": 100, "-SOS.View": 2, "-SOS.View.Dphi": 120, "-AER.Model": 0, "-AER.MMD.SDtype": 1, "-AER.MMD.LNDradius": 0.3,
        "-AER.MMD.LNDvar": 0.6, "-AER.MMD.MRwa": 1.45, "-AER.MMD.MIwa": -0.003, "-AER.MMD.MRwaref": 1.45, "-AER.MMD.MIwaref": -0.003,
        "-ANG.Rad.NbGauss": 16, "-ANG.Aer.NbGauss": 20, "-AP.AbsProfile.Type": 2, "-AP.CO2": 420.0, "-AP.CH4": 1.9,
        "-AER.AOTref": 0.15, "-AER.Tronca": 1, "-SURF.Type": 7, "-SURF.Alb": 0.02, "-SURF.Ind": 1.5, "-SURF.Maignan.C": 4.0,
        "-SURF.Roujean.K0": 0.2, "-SURF.Roujean.K1": 0.03, "-SURF.Roujean.K2": 0.25, "-SOS_Main.Log": "NO_LOG_FILE",
        "-SOS.Flux": "NO_OUTPUT"}


def spectrum_kwargs(rs, every=1):
    kws = []
    for nu in np.arange(27495.0, 2500.0, -10.0)[::every]:
        wa = 1.0e4 / nu
        if wa < 0.3641 or wa > 3.999:
            continue
        u = dict(USER)
        u["-SOS_Main.Wa"] = float(wa)
        kws.append(rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), u), trace=False))
    return kws


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=1, help="take every k-th interval of the spectrum")
    ap.add_argument("--loop", type=int, default=120, help="wavelengths timed in the plain sos_proc loop / sos_proc_many")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--streams", type=int, default=16, help="side streams of sos_spectrum (prep_streams)")
    ap.add_argument("--pool", type=int, default=0, help="also run the spectrum through spectrum_pool with this many host processes")
    ap.add_argument("--pool-only", action="store_true", help="only the spectrum_pool run: this process never touches the GPU")
    a = ap.parse_args()
    if a.pool_only:
        pkg = importlib.import_module("radiativetransfer-sos_amd")
        rs, sp = pkg.run_sos, pkg.spectrum_pool
        root = tempfile.mkdtemp(prefix="synth_fic_")
        synth_ckd.write_tables(root)
        os.environ["SOS_ABS_ROOT"] = root
        kws = spectrum_kwargs(rs, a.every)
        nb = sum(pkg.absorption.band_bin_count(kw["wa_simu"], 10.0) for kw in kws)
        with sp.SpectrumPool(processes=a.pool or 4) as pool:
            pool.run(kws[::8])
            for _ in range(2):
                t0 = time.perf_counter()
                pool.run(kws)
                dt = time.perf_counter() - t0
                print("(d) spectrum_pool alone, %d processes: %4d wavelengths (%d bins) in %6.2f s = %7.1f wavelengths/s" % (
                    a.pool or 4, len(kws), nb, dt, len(kws) / dt), flush=True)
        return
    import torch
    pkg = importlib.import_module("radiativetransfer-sos_amd")
    rs = pkg.run_sos
    root = tempfile.mkdtemp(prefix="synth_fic_")
    t0 = time.perf_counter()
    nfile = synth_ckd.write_tables(root)
    os.environ["SOS_ABS_ROOT"] = root
    kws = spectrum_kwargs(rs, a.every)
    nb = sum(pkg.absorption.band_bin_count(kw["wa_simu"], 10.0) for kw in kws)
    print("spectrum: %d wavelengths, %d CKD bins (tables: %d files written and parsed in %.1f s)" % (
        len(kws), nb, nfile, time.perf_counter() - t0), flush=True)
    sub = kws[::max(1, len(kws) // a.loop)][:a.loop]
    nbs = sum(pkg.absorption.band_bin_count(kw["wa_simu"], 10.0) for kw in sub)
    rs.sos_proc(**sub[0]); torch.cuda.synchronize()                      # warm-up: library load, surface matrices, caches
    t0 = time.perf_counter()
    seq = [rs.sos_proc(**kw) for kw in sub]
    dt = time.perf_counter() - t0
    print("(a) plain sos_proc loop     : %4d wavelengths (%d bins) in %6.2f s = %7.1f calls/s, %8.1f bins/s" % (len(sub), nbs, dt, len(sub) / dt, nbs / dt), flush=True)
    t0 = time.perf_counter()
    many = rs.sos_proc_many(sub, n_workers=8)
    dt = time.perf_counter() - t0
    print("(b) sos_proc_many, 8 threads: %4d wavelengths (%d bins) in %6.2f s = %7.1f calls/s, %8.1f bins/s" % (len(sub), nbs, dt, len(sub) / dt, nbs / dt), flush=True)
    rs.sos_spectrum(sub[:8])
    tm = {}
    t0 = time.perf_counter()
    spec = rs.sos_spectrum(sub, timings=tm, chunk=a.chunk)
    dt = time.perf_counter() - t0
    print("(c) sos_spectrum, same list : %4d wavelengths (%d bins) in %6.2f s = %7.1f calls/s, %8.1f bins/s" % (len(sub), nbs, dt, len(sub) / dt, nbs / dt), flush=True)
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for s1, s2 in zip(seq, spec) for x, y in zip(s1, s2))
    same_many = all(np.array_equal(np.asarray(x), np.asarray(y)) for s1, s2 in zip(seq, many) for x, y in zip(s1, s2))
    print("    outputs identical to the plain loop, bit for bit: sos_spectrum %s, sos_proc_many %s" % (same, same_many), flush=True)
    if rs.PREPARE_SEGMENTS is not None:
        rs.PREPARE_SEGMENTS.clear()
    tm = {}
    t0 = time.perf_counter()
    full = rs.sos_spectrum(kws, timings=tm, chunk=a.chunk, prep_streams=a.streams)
    dt = time.perf_counter() - t0
    print("(c) sos_spectrum, FULL      : %4d wavelengths (%d bins) in %6.2f s = %7.1f wavelengths/s, %8.1f bins/s" % (len(kws), nb, dt, len(kws) / dt, nb / dt))
    print("    host phases per wavelength (ms): " + ", ".join("%s %.3f" % (k, 1e3 * v / len(kws)) for k, v in tm.items()), flush=True)
    tm2 = {}
    t0 = time.perf_counter()
    rs.sos_spectrum(kws, timings=tm2, chunk=a.chunk, prep_streams=a.streams)
    dt2 = time.perf_counter() - t0
    print("(c) sos_spectrum, FULL again: %4d wavelengths (%d bins) in %6.2f s = %7.1f wavelengths/s   (every table file parsed by now)" % (
        len(kws), nb, dt2, len(kws) / dt2))
    print("    host phases per wavelength (ms): " + ", ".join("%s %.3f" % (k, 1e3 * v / len(kws)) for k, v in tm2.items()), flush=True)
    if a.pool:
        sp = pkg.spectrum_pool
        t0 = time.perf_counter()
        pool = sp.SpectrumPool(processes=a.pool)
        t1 = time.perf_counter()
        pool.run(kws[::8])                                   # warm-up of the workers: table parsing, Mie records, surface matrices
        t2 = time.perf_counter()
        outs = pool.run(kws)
        dt = time.perf_counter() - t2
        print("(d) spectrum_pool, %d processes: %4d wavelengths (%d bins) in %6.2f s = %7.1f wavelengths/s, %8.1f bins/s   (pool start %.1f s, "
              "warm-up run %.1f s)" % (a.pool, len(kws), nb, dt, len(kws) / dt, nb / dt, t1 - t0, t2 - t1), flush=True)
        same = all(np.array_equal(np.asarray(x), np.asarray(y)) for s1, s2 in zip(full, outs) for x, y in zip(s1, s2))
        print("    outputs identical to sos_spectrum in this process, bit for bit: %s" % same, flush=True)
        pool.close()
    if rs.PREPARE_SEGMENTS:                                  # SOS_PREPARE_SEGMENTS=1: where the preparation spends its host time
        print("    prepare, by segment (ms)       : " + ", ".join("%s %.3f" % (k, 1e3 * v / len(kws)) for k, v in rs.PREPARE_SEGMENTS.items()),
              flush=True)
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        rs.sos_spectrum(kws[:400], chunk=a.chunk)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue())


if __name__ == "__main__":
    main()
