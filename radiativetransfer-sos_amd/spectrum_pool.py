"""Several host processes behind one sos_spectrum call.

The preparation of a wavelength is ~0.6 ms of Python under one interpreter lock against ~0.1 ms of GPU time
(profiles/r03_hyperspectral.txt), so one interpreter per GPU leaves the card idle.  `SpectrumPool` starts a few worker
processes on the SAME GPU (each its own interpreter, HIP context and few hardware queues -- a card serves about sixteen
queues in all, bench.py), deals the wavelengths of a spectrum to them by cost (run_sos.spectrum_costs,
dist.balanced_shards: the partition of the multi-GPU path) and collects the 23-tuples; every tuple is what
run_sos.sos_spectrum returns for that call, bit for bit (the workers run exactly that function on their share).

    pool = SpectrumPool(processes=4)            # workers import the package and initialise the GPU once
    outs = pool.run(kwargs_list)                # as run_sos.sos_spectrum(kwargs_list)
    pool.close()
or  outs = sos_spectrum_processes(kwargs_list, processes=4)      # a module-level pool, kept for the next call

Measured (scripts/hyperspectral_bench.py --pool 4, 2496 wavelengths of BASELINE config 5): 2105 wavelengths/s from a caller
that does not use the GPU itself, against 1450-1500 for sos_spectrum in one process.  The caller's own HIP context counts
towards the card's queues: from a process that has already run GPU work on 16 hardware queues (GPU_MAX_HW_QUEUES=16) the
same pool gives 1500 -- start such a caller with a small GPU_MAX_HW_QUEUES, or let the pool do all the GPU work.

The workers are plain child processes speaking length-prefixed pickles over their stdin / stdout (no multiprocessing start
method, so nothing is re-imported from the caller's __main__).  Under `torchrun` use run_sos.sos_spectrum itself: the ranks
are the processes.  Result files of a call (-SOS_Main.ResRoot) are written by the worker that owns it.
"""
import atexit
import os
import pickle
import struct
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _send(f, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(b)))
    f.write(b)
    f.flush()


def _recv(f):
    h = f.read(8)
    if len(h) < 8:
        raise EOFError("spectrum worker closed its pipe")
    n = struct.unpack("<Q", h)[0]
    b = f.read(n)
    if len(b) < n:
        raise EOFError("spectrum worker closed its pipe")
    return pickle.loads(b)


def _rows_of(kw):
    """Azimuth rows in use in the (361,81) tables of the call (run_sos._trphi_azimuths)."""
    return 2 if int(kw["itrphi"]) == 1 else len(range(0, 361, int(kw["pas_phi"])))


class SpectrumPool:
    def __init__(self, processes=4, device=0, queues=None):
        self.n = max(1, int(processes))
        self.device = int(device)
        env = dict(os.environ)
        env["GPU_MAX_HW_QUEUES"] = str(int(queues) if queues else max(2, 16 // self.n))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):     # a worker is not a rank of the caller's group
            env.pop(k, None)
        self.workers = []
        for _ in range(self.n):
            self.workers.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--worker", str(self.device)],
                                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env))
        try:
            for w in self.workers:                        # each answers once its package is imported and its GPU initialised
                r = _recv(w.stdout)
                if r != "ready":
                    raise RuntimeError("spectrum worker failed to start: %r" % (r,))
        except BaseException:
            self.close()
            raise

    def run(self, kwargs_list, aer_phases=None):
        """run_sos.sos_spectrum(kwargs_list, aer_phases) with the wavelengths dealt to the pool's processes."""
        import importlib
        pkg = importlib.import_module(os.path.basename(_HERE))
        rs, dist = pkg.run_sos, pkg.dist
        nwl = len(kwargs_list)
        if aer_phases is None:
            aer_phases = [None] * nwl
        shards = dist.balanced_shards(rs.spectrum_costs(kwargs_list), self.n)
        jobs = []
        for w, sh in zip(self.workers, shards):
            idx = [int(i) for i in sh]
            if idx:
                _send(w.stdin, ("spectrum", [kwargs_list[i] for i in idx], [aer_phases[i] for i in idx],
                                {k: os.environ.get(k) for k in ("SOS_ABS_ROOT",)}))     # (the data root may change between runs)
                jobs.append((w, idx))
        results = [None] * nwl
        error = None
        replies = []
        for w, idx in jobs:                               # (every worker has its job already: they run side by side)
            replies.append((idx, _recv(w.stdout)))
        nall = sum(len(idx) for idx, r in replies if r[0] == "ok")
        blocks = rs._zero_pages((nall, len(rs._TABLE_NAMES), 361, 81)) if nall else None
        k = 0
        for idx, r in replies:
            if r[0] != "ok":
                error = error or r
                continue
            for i, c in zip(idx, r[1]):
                results[i] = rs._expand_outputs(c, blocks[k])
                k += 1
        if error is not None:
            e = rs.SosProcError(error[1], ier=error[2]) if error[0] == "sos" else RuntimeError(error[1])
            raise e
        return results

    def close(self):
        for w in getattr(self, "workers", []):
            try:
                if w.poll() is None:
                    _send(w.stdin, ("quit",))
                    w.stdin.close()
            except Exception:
                pass
        for w in getattr(self, "workers", []):
            try:
                w.wait(timeout=20)
            except Exception:
                w.kill()
        self.workers = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


_POOL = None


def sos_spectrum_processes(kwargs_list, aer_phases=None, processes=4, device=0):
    """run_sos.sos_spectrum through a module-level pool of `processes` host processes on GPU `device` (started at the first
    call: a few seconds of imports and GPU initialisation per worker; kept for later calls, closed at interpreter exit)."""
    global _POOL
    if _POOL is not None and (_POOL.n != max(1, int(processes)) or _POOL.device != int(device) or
                              any(w.poll() is not None for w in _POOL.workers)):
        _POOL.close()
        _POOL = None
    if _POOL is None:
        _POOL = SpectrumPool(processes, device)
        atexit.register(close_pool)
    return _POOL.run(kwargs_list, aer_phases)


def close_pool():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL = None


def _worker(device):
    import importlib
    out = sys.stdout.buffer                               # the protocol's pipe; anything printed goes to stderr from here on
    sys.stdout = sys.stderr
    try:
        sys.path.insert(0, os.path.dirname(_HERE))
        pkg = importlib.import_module(os.path.basename(_HERE))
        import torch
        torch.cuda.set_device(device)
        torch.zeros(1, device="cuda")
        pkg.capi.lib()
        rs = pkg.run_sos
    except BaseException as e:
        _send(out, ("failed", repr(e)))
        return 1
    _send(out, "ready")
    inp = sys.stdin.buffer
    while True:
        try:
            job = _recv(inp)
        except EOFError:
            return 0
        if job[0] == "quit":
            return 0
        _, kws, aers, env = job
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        try:
            res = rs.sos_spectrum(kws, aer_phases=aers, device=device, gather=False)
            _send(out, ("ok", [rs._compact_outputs(t, _rows_of(kw)) for t, kw in zip(res, kws)]))
        except rs.SosProcError as e:
            _send(out, ("sos", str(e), getattr(e, "ier", 1)))
        except BaseException as e:
            _send(out, ("error", "%s: %s" % (type(e).__name__, e), 1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--worker":
        sys.exit(_worker(int(sys.argv[2])))
    sys.exit("spectrum_pool.py is a library module (workers are started by SpectrumPool)")
