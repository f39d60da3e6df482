"""Host-side driver of the MI355X hot path: one `SosContext` per wavelength, batches of CKD bins solved
by the fused HIP kernel, aggregation on device.  torch is used for device memory and streams only; all
arithmetic happens in libsosgpu.so (capi.py), and there is no CPU fallback.

Mirrors the reference per-wavelength sequence of SOS_PROC (src/SOS_PROC.F:3423-3594):
  SOS_PREPA_OS -> [per bin: SOS (truncation rescale) -> SOS_OS -> SOS_AGGREGATE].
"""
import collections
import ctypes as C
import threading

import numpy as np
import torch

from . import capi
from .synth import MDF_DEFAULT


class SosBinError(RuntimeError):
    """A bin of the batch is malformed / needs more than CTE_OS_NT levels: the reference's IER = -1
    (SOS_OS.F:1627-1648, SOS_PROFIL.F)."""
    ier = -1


def _dev_f64(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float64).contiguous()
    a = np.ascontiguousarray(x, dtype=np.float64)
    if not a.flags.writeable:                      # (cached read-only tables: torch wants a writable buffer to wrap)
        a = a.copy()
    return _upload(torch.from_numpy(a), device)


def _upload(t, device):
    """Host tensor -> device on the current stream without a host wait: through a pinned staging block of torch's host
    allocator (returned to it when the copy has passed).  A pageable `.to(device)` waits for the stream -- behind whatever
    preparation kernels sos_spectrum has queued there."""
    if t.numel() == 0 or t.numel() * t.element_size() > (4 << 20) or torch.device(device).type != "cuda":
        return t.to(device)                        # (large tables: uploaded once per batch, not worth a pinned block)
    return t.pin_memory().to(device, non_blocking=True)


def _dev_i32(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int32).contiguous()
    return _upload(torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)), device)


_CONST_DEV = collections.OrderedDict()
_CONST_LOCK = threading.Lock()


def _const_dev_f64(x, device):
    """Device copy of a SMALL host array that repeats from call to call (altitude grid of the gas profiles, azimuth list of a
    view): kept per (content, device) -- a pageable upload is a host-synchronous 30 us, a spectrum makes thousands of them."""
    a = np.ascontiguousarray(x, dtype=np.float64)
    if a.nbytes > 8192:
        return _dev_f64(a, device)
    key = (a.tobytes(), str(device))
    with _CONST_LOCK:
        hit = _CONST_DEV.get(key)
        if hit is not None:
            _CONST_DEV.move_to_end(key)
            return hit
    t = torch.from_numpy(a.copy()).to(device)              # (synchronous copy: complete, usable from any stream)
    with _CONST_LOCK:
        _CONST_DEV[key] = t
        while len(_CONST_DEV) > 64:
            _CONST_DEV.popitem(last=False)
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def nogas_profile(tr, hr, ta, ha, device=0):
    """Queue the no-gas profile of a wavelength (SOS_PROFIL.F:349-489) on the current stream ahead of make_profiles
    (sosgpu_profile_nogas): its level placement is a ~1 ms serial chain on one wavefront that only needs the two optical
    thicknesses and scale heights, so it can run while the host is still preparing the wavelength.  Returns the device block
    to pass to SosContext.make_profiles(nogas=...)."""
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    ng = torch.empty(4 * capi.NOGAS_LEVELS, dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    capi.check(capi.lib().sosgpu_profile_nogas(dev.index or 0, float(tr), float(hr), float(ta), float(ha), _ptr(ng), st),
               "sosgpu_profile_nogas")
    return ng


SEG = None          # run_sos._seg when SOS_PREPARE_SEGMENTS is set (host-time diagnostic), else None


class SosContext:
    """Everything SOS_OS needs that does not depend on the CKD bin (angles, phase-matrix expansion,
    surface), resident on one GPU, with the Fourier kernels of every order precomputed
    (sosgpu_noyaux, replaces SOS_NOYAUX SOS_OS.F:1857)."""

    def __init__(self, mu, ga, n0, alpha, beta, gamma, zeta, *, iborm_max=None, ro=0.0, imat_surf=0,
                 ifresnel=0, ind_surf=1.34, ron=MDF_DEFAULT, ipolar=1, igmax=100, rsurf=None, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("SosContext needs a GPU (gfx950); there is no CPU fallback in the product path")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.n = int(len(mu))
        self.w = 2 * self.n + 1
        self.os_nb = int(len(beta) - 1)
        self.smax = self.os_nb if iborm_max is None else int(iborm_max)
        self.n0 = int(n0)
        self.mu = np.ascontiguousarray(mu, dtype=np.float64)
        self.ga = np.ascontiguousarray(ga, dtype=np.float64)
        coefs = [np.ascontiguousarray(x, dtype=np.float64) for x in (alpha, beta, gamma, zeta)]
        self._coefs = coefs
        self._opts = dict(ind_surf=float(ind_surf), ron=float(ron), ipolar=int(ipolar), igmax=int(igmax))
        for x in coefs:
            if len(x) != self.os_nb + 1:
                raise ValueError("alpha/beta/gamma/zeta must all have os_nb+1 entries")
        wv = capi.Wave(n=self.n, os_nb=self.os_nb, n0=self.n0, imat_surf=int(imat_surf), ifresnel=int(ifresnel),
                       ipolar=int(ipolar), igmax=int(igmax), reserved=0, ro=float(ro), ind_surf=float(ind_surf),
                       ron=float(ron))
        self._h = C.c_void_p()
        L = capi.lib()
        dp = lambda a: a.ctypes.data_as(C.c_void_p)
        SEG and SEG("context: python")
        capi.check(L.sosgpu_create(C.byref(self._h), self.device.index or 0, C.byref(wv), dp(self.mu), dp(self.ga),
                                   dp(coefs[0]), dp(coefs[1]), dp(coefs[2]), dp(coefs[3]), self.smax),
                   "sosgpu_create")
        SEG and SEG("context: sosgpu_create")
        self._rsurf = None
        if int(imat_surf) == 1:
            if rsurf is None:
                raise ValueError("imat_surf=1 needs rsurf[smax+1][9][N][N] (float32)")
            if isinstance(rsurf, torch.Tensor):
                r = rsurf.to(device=self.device, dtype=torch.float32).contiguous()
            else:
                r = torch.from_numpy(np.ascontiguousarray(rsurf, dtype=np.float32)).to(self.device)
            if tuple(r.shape) != (self.smax + 1, 9, self.n, self.n):
                raise ValueError("rsurf shape %s != %s" % (tuple(r.shape), (self.smax + 1, 9, self.n, self.n)))
            self._rsurf = r               # kept alive: the packing below is only queued
            capi.check(L.sosgpu_set_surface_matrices_async(self._h, _ptr(r), self._stream()),
                       "sosgpu_set_surface_matrices_async")
            SEG and SEG("context: surface matrices")
        self.noyaux()
        SEG and SEG("context: sosgpu_noyaux")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def noyaux(self):
        capi.check(capi.lib().sosgpu_noyaux(self._h, self._stream()), "sosgpu_noyaux")

    def noyaux_fetch(self, is_):
        """Kernels of Fourier order is_ in the reference layout (parity accessor)."""
        w = self.w
        out = np.zeros(6 * w * w + 3 * w)
        torch.cuda.synchronize(self.device)
        capi.check(capi.lib().sosgpu_noyaux_fetch(self._h, int(is_), out.ctypes.data_as(C.c_void_p)), "sosgpu_noyaux_fetch")
        names = ["BP", "GR", "GT", "ARR", "ART", "ATT"]
        d = {k: out[i * w * w:(i + 1) * w * w].reshape(w, w) for i, k in enumerate(names)}
        for i, k in enumerate(["XPL", "XRL", "XTL"]):
            d[k] = out[6 * w * w + i * w:6 * w * w + (i + 1) * w]
        return d

    def upload_bins(self, h, xdel, ydel, nt=None, iborm=None, zout=-1.0, zprof=None, order=None):
        """Pack per-bin profiles (after the SOS.F rescale) into the device layout of sosgpu_os_solve.
        h/xdel/ydel: [nb][L] arrays (ragged bins: pass nt[nb] and pad).
        order="cost": bins are permuted by decreasing total optical depth before upload (the returned dict holds
        `perm`, with result[i] belonging to input bin perm[i]).  Bins of similar cost then run side by side, drift
        less across Fourier orders and keep re-reading the same source operators from L2 (scheduling only: every
        bin's result is unchanged)."""
        h = np.atleast_2d(np.asarray(h, dtype=np.float64))
        xdel = np.atleast_2d(np.asarray(xdel, dtype=np.float64))
        ydel = np.atleast_2d(np.asarray(ydel, dtype=np.float64))
        nb, lmax = h.shape
        nt = np.full(nb, lmax - 1, dtype=np.int32) if nt is None else np.asarray(nt, dtype=np.int32)
        perm = None
        if order == "cost":
            perm = np.argsort(-h[np.arange(nb), nt], kind="stable")
            h, xdel, ydel, nt = h[perm], xdel[perm], ydel[perm], nt[perm]
            if iborm is not None and np.ndim(iborm):
                iborm = np.asarray(iborm)[perm]
            if zprof is not None:
                zprof = np.atleast_2d(np.asarray(zprof, dtype=np.float64))[perm]
        lp = int(-(-lmax // 16) * 16)
        prof = np.zeros((nb, 3, lp))
        prof[:, 0, :lmax], prof[:, 1, :lmax], prof[:, 2, :lmax] = h, xdel, ydel
        if iborm is None:
            # SOS.F:549-550: IBORM = 2 for a purely molecular bin, OS_NB otherwise
            iborm = np.where(np.any(xdel != 0.0, axis=1), self.smax, min(2, self.smax)).astype(np.int32)
        iborm = np.broadcast_to(np.asarray(iborm, dtype=np.int32), (nb,)).copy()
        jout = zz = None
        if zout != -1.0:
            # SOS_OS.F:1514-1520: first level J with ZOUT >= ZPROF(J), linear weight ZZ
            zprof = np.atleast_2d(np.asarray(zprof, dtype=np.float64))
            jout = np.zeros(nb, dtype=np.int32)
            zz = np.zeros(nb)
            for b in range(nb):
                j = 1
                while zout < zprof[b, j]:
                    j += 1
                jout[b] = j
                zz[b] = (zout - zprof[b, j - 1]) / (zprof[b, j] - zprof[b, j - 1])
        d = self.device
        return dict(nb=nb, lp=lp, perm=perm, nt=_dev_i32(nt, d), iborm=_dev_i32(iborm, d), prof=_dev_f64(prof, d),
                    jout=None if jout is None else _dev_i32(jout, d), zz=None if zz is None else _dev_f64(zz, d))

    def absorption_profiles(self, ik, xk, ro):
        """SOS_ABSPROFILE (SOS_ABSPROFILE.F:325-371) of every CKD bin on the device (sosgpu_absprofile): ik[nb][8] 1-based
        exponential term per gas, xk[8][nterm][nlev-1] / ro[8][nlev-1] from absorption.layer_tables.  Returns the device
        tensor tabs[nb][nlev] that make_profiles takes."""
        d = self.device
        if isinstance(ik, torch.Tensor) or isinstance(xk, torch.Tensor) or isinstance(ro, torch.Tensor):
            ik_t = _dev_i32(ik, d)
            xk_t, ro_t = _dev_f64(xk, d), _dev_f64(ro, d)
        else:
            # one upload for the three tables (8-byte units; a bin's eight int32 indices are four of them)
            ik_h = np.ascontiguousarray(ik, dtype=np.int32)
            xk_h, ro_h = np.ascontiguousarray(xk, dtype=np.float64), np.ascontiguousarray(ro, dtype=np.float64)
            if ik_h.ndim != 2 or ik_h.shape[1] != 8:
                raise ValueError("ik must be [nb][8], xk [8][nterm][nlev-1], ro [8][nlev-1]")
            buf = _upload(torch.from_numpy(np.concatenate([xk_h.ravel(), ro_h.ravel(), ik_h.ravel().view(np.float64)])), d)
            xk_t = buf[:xk_h.size].view(xk_h.shape)
            ro_t = buf[xk_h.size:xk_h.size + ro_h.size].view(ro_h.shape)
            ik_t = buf[xk_h.size + ro_h.size:].view(torch.int32).view(ik_h.shape)
        nb, nterm, nlev = int(ik_t.shape[0]), int(xk_t.shape[1]), int(xk_t.shape[2]) + 1
        if tuple(ik_t.shape) != (nb, 8) or xk_t.shape[0] != 8 or tuple(ro_t.shape) != (8, nlev - 1):
            raise ValueError("ik must be [nb][8], xk [8][nterm][nlev-1], ro [8][nlev-1]")
        tabs = torch.empty((nb, nlev), dtype=torch.float64, device=d)
        capi.check(capi.lib().sosgpu_absprofile(d.index or 0, nb, nlev, nterm, _ptr(ik_t), _ptr(xk_t), _ptr(ro_t), _ptr(tabs),
                                                self._stream()), "sosgpu_absprofile")
        return tabs

    def make_profiles(self, nb, tr, hr, ta, ha, altabs=None, tabs=None, *, a_tronc=0.0, piz=1.0, piztr=1.0, zout=-1.0,
                      lp=608, absprofil=1, nogas=None):
        """SOS_PROFILE (IPROFIL=1) + SOS_DISC + the SOS.F rescale for nb CKD bins ON THE DEVICE (sosgpu_profile):
        tabs[nb][nblev] is each bin's cumulative gas absorption optical depth on the descending altitude grid
        altabs[nblev] (None: no gas).  Returns the same dict upload_bins returns (ready for solve()), plus `zprof` and
        the per-bin scalars `scal` [nb][4] = {0, TTOT_TRONC, TTOT_VRAI, TAUOUT} that aggregate() takes.
        Bins whose profile needs more than CTE_OS_NT levels come back with nt = -1 (the reference's IER = -1).
        nogas: the block nogas_profile(tr, hr, ta, ha) queued earlier on this stream (head start), or None."""
        d = self.device
        t_alt = t_tab = None
        nblev = 0
        if tabs is not None:
            t_tab = tabs if isinstance(tabs, torch.Tensor) else np.atleast_2d(np.asarray(tabs, dtype=np.float64))
            t_tab = _dev_f64(t_tab, d)
            t_alt = _const_dev_f64(altabs, d)
            nblev = int(t_alt.numel())
            if t_tab.shape != (nb, nblev):
                raise ValueError("tabs must be [nb][len(altabs)]")
        # (two cleared blocks instead of seven: a band has 1-125 bins, each fill is a launch)
        fb = torch.zeros(nb * (4 * lp + 5), dtype=torch.float64, device=d)
        ib = torch.zeros(3 * nb, dtype=torch.int32, device=d)
        prof = fb[:nb * 3 * lp].view(nb, 3, lp)
        zprof = fb[nb * 3 * lp:nb * 4 * lp].view(nb, lp)
        scal = fb[nb * 4 * lp:nb * (4 * lp + 4)].view(nb, 4)
        nt, iborm = ib[:nb], ib[nb:2 * nb]
        jout = zz = None
        if zout != -1.0:
            jout = ib[2 * nb:]
            zz = fb[nb * (4 * lp + 4):]
        capi.check(capi.lib().sosgpu_profile(self._h, nb, tr, hr, ta, ha, int(absprofil), nblev, _ptr(t_alt), _ptr(t_tab),
                                             a_tronc, piz, piztr, zout, lp, _ptr(prof), _ptr(nt), _ptr(iborm),
                                             _ptr(zprof), _ptr(jout), _ptr(zz), _ptr(scal), _ptr(nogas), self._stream()),
                   "sosgpu_profile")
        return dict(nb=nb, lp=lp, perm=None, nt=nt, iborm=iborm, prof=prof, jout=jout, zz=zz, zprof=zprof, scal=scal)

    def solve_band(self, bins, aik, seg=None, group=None, tdifmug=None, reduce=True):
        """The whole per-wavelength bin loop of SOS_PROC (SOS_PROC.F:3459-3594) for bins already on the device
        (upload_bins / make_profiles): fused SOS_OS of every bin, AIK-weighted SOS_AGGREGATE, and -- when
        torch.distributed is initialised with more than one rank -- the one all-reduce that joins the ranks' bin
        slices.  aik[nb]: this rank's normalised weights (in the order of `bins`; apply bins["perm"] first when the
        upload was cost-sorted).  A rank may hold no bin at all (nb = 0): it contributes the neutral element.
        reduce=False: this rank holds the whole band (a replica, not a shard) -- no collective.
        Raises SosBinError (the reference's IER = -1) on every rank when any bin of the band is malformed.
        Returns (rec[nseg][smax+1][3][W] device tensor, scalars dict of dist.finish_scalars)."""
        from . import dist as _dist
        out = self.solve(bins) if bins["nb"] else None
        rec, scal = self.aggregate(out, aik, seg=seg, scal=bins.get("scal"), tdifmug=tdifmug)
        if reduce:
            buf = _dist.all_reduce_partial(_dist.pack_partial(rec, scal), scal.shape[1], group=group)
            rec, scal = _dist.unpack_partial(buf, rec.shape)
        fin = _dist.finish_scalars(scal)
        if (fin["min_orders"] < 0).any():
            raise SosBinError("SOS_OS: %d band(s) hold a malformed bin (NT outside 1..CTE_OS_NT or IBORM out of range)"
                              % int((fin["min_orders"] < 0).sum()))
        return rec, fin

    def diffuse_transmissions(self, bins):
        """Diffuse transmissions of the `-SOS.Trans` option (SOS.F:600-635): for the solar direction and for every
        direction J as incidence, SOS_OS restricted to Fourier order 0 over a black, non-reflecting ground
        (RHO = 0, IMAT_SURF = IFRESNEL = 0, IBORM = 0, ZOUT = -1); its EMOINS is TDIFMUS (N0 = solar index) resp.
        TDIFMUG(J) (N0 = J).  One order-0 context per incidence direction (they differ in the single-scattering
        vectors only), all bins per launch.  Returns (tdifmus[nb], tdifmug[nb][N]) as device tensors; values are for
        the equivalent (truncated) atmosphere, as SOS_OS delivers them."""
        nb = bins["nb"]
        zeros = torch.zeros(nb, dtype=torch.int32, device=self.device)
        sub = dict(bins, iborm=zeros, jout=None, zz=None)
        tdifmug = torch.empty((nb, self.n), dtype=torch.float64, device=self.device)
        tdifmus = None
        for j in range(1, self.n + 1):
            cx = SosContext(self.mu, self.ga, j, *self._coefs, iborm_max=0, ro=0.0, imat_surf=0, ifresnel=0,
                            device=self.device, **self._opts)
            try:
                out = cx.solve(sub)
                tdifmug[:, j - 1] = out["flux"][:, 0]
                if j == self.n0:
                    tdifmus = out["flux"][:, 0].clone()
                torch.cuda.synchronize(self.device)
            finally:
                cx.close()
        return tdifmus, tdifmug

    def alloc_outputs(self, nb, zero=True):
        """Output buffers of solve().  zero=True: rec is zero-initialised (the kernel only writes the orders a bin runs,
        sosgpu.h); zero=False: plain allocations -- for callers that read the records through norders only (aggregate does),
        it saves four fill kernels per call (they matter when many small launches share the GPU, solve_many)."""
        d = self.device
        mk = torch.zeros if zero else torch.empty
        return dict(rec=mk((nb, self.smax + 1, 3, self.w), dtype=torch.float64, device=d),
                    norders=mk(nb, dtype=torch.int32, device=d),
                    iglast=mk((nb, self.smax + 1), dtype=torch.int32, device=d),
                    flux=mk((nb, 2), dtype=torch.float64, device=d))

    def solve(self, bins, out=None):
        """Run the fused SOS_OS kernel on a batch of bins already resident in HBM (upload_bins)."""
        if out is None:
            out = self.alloc_outputs(bins["nb"])
        capi.check(capi.lib().sosgpu_os_solve(self._h, bins["nb"], bins["lp"], _ptr(bins["nt"]), _ptr(bins["iborm"]),
                                              _ptr(bins["prof"]), _ptr(bins["jout"]), _ptr(bins["zz"]),
                                              _ptr(out["rec"]), _ptr(out["norders"]), _ptr(out["iglast"]),
                                              _ptr(out["flux"]), self._stream()), "sosgpu_os_solve")
        return out

    def last_solve_ms(self):
        ms = C.c_float(0)
        capi.check(capi.lib().sosgpu_last_solve_ms(self._h, C.byref(ms)), "sosgpu_last_solve_ms")
        return ms.value

    def solve_flops(self, bins, out):
        """(reference-algorithm flops, flops of the parity / rank-4 form executed) of the last solve; see
        sosgpu_os_flops in include/sosgpu.h."""
        fl = (C.c_double * 2)(0.0, 0.0)
        torch.cuda.synchronize(self.device)
        capi.check(capi.lib().sosgpu_os_flops(self._h, bins["nb"], _ptr(bins["nt"]), _ptr(out["norders"]),
                                              _ptr(out["iglast"]), fl), "sosgpu_os_flops")
        return fl[0], fl[1]

    def aggregate(self, out, aik, seg=None, scal=None, tdifmug=None):
        """SOS_AGGREGATE over segments of bins (default: all bins = one wavelength).  out = None or an empty batch
        gives the neutral element (a rank without bins).  tdifmug: optional per-bin [nb][N] diffuse transmissions.
        Returns (rec[nseg][smax+1][3][W], scal[nseg][10+N]) on device; see include/sosgpu.h for scal."""
        nb = 0 if out is None else out["rec"].shape[0]
        d = self.device
        sw = capi.SCAL_BASE + self.n
        if nb == 0:
            o_rec = torch.empty((1, self.smax + 1, 3, self.w), dtype=torch.float64, device=d)
            o_scal = torch.empty((1, sw), dtype=torch.float64, device=d)
            capi.check(capi.lib().sosgpu_aggregate(self._h, 0, 1, None, None, None, None, None, None, None,
                                                   _ptr(o_rec), _ptr(o_scal), self._stream()), "sosgpu_aggregate")
            return o_rec, o_scal
        if seg is None:
            # the default segment table [0, nb] is kept on the device: a host-to-device copy here would block the host until the
            # solve queued before it on this stream has finished, and with it every overlap of wavelengths on other streams
            cache = self.__dict__.setdefault("_seg_cache", {})
            seg_t = cache.get(nb)
            if seg_t is None:
                seg_t = cache[nb] = _dev_i32(np.array([0, nb], dtype=np.int32), d)
        else:
            seg_t = _dev_i32(seg, d)
        nseg = seg_t.numel() - 1
        aik_t = _dev_f64(aik, d)
        scal_t = torch.zeros((nb, 4), dtype=torch.float64, device=d) if scal is None else _dev_f64(scal, d)
        tdg = None if tdifmug is None else _dev_f64(tdifmug, d)
        if tdg is not None and tuple(tdg.shape) != (nb, self.n):
            raise ValueError("tdifmug must be [nb][N]")
        o_rec = torch.empty((nseg, self.smax + 1, 3, self.w), dtype=torch.float64, device=d)
        o_scal = torch.empty((nseg, sw), dtype=torch.float64, device=d)
        capi.check(capi.lib().sosgpu_aggregate(self._h, nb, nseg, _ptr(seg_t), _ptr(aik_t), _ptr(out["rec"]),
                                               _ptr(out["norders"]), _ptr(out["flux"]), _ptr(scal_t), _ptr(tdg),
                                               _ptr(o_rec), _ptr(o_scal), self._stream()), "sosgpu_aggregate")
        return o_rec, o_scal

    def trphi(self, rec, nf, tau, tauout, phis_rad, igli=0, wind=0.0, land=None):
        """SOS_TRPHI + SOS_POLAR for a list of azimuths (radians).  rec: device tensor [>=nf][3][W]
        (aggregated records).  Returns a device tensor [nphi][7][W]: XIT, XQT, XUT, ANGDIFF, polarisation
        angle, polarisation rate, polarised radiance."""
        d = self.device
        phis = _const_dev_f64(np.atleast_1d(np.asarray(phis_rad, dtype=np.float64)), d)
        rec = rec.to(device=d, dtype=torch.float64).contiguous()
        out = torch.empty((phis.numel(), 7, self.w), dtype=torch.float64, device=d)
        lp = None if land is None else C.byref(land)          # capi.Land: direct term of the land model (-SURF.Type 3..7)
        capi.check(capi.lib().sosgpu_trphi(self._h, int(nf), _ptr(rec), float(tau), float(tauout), phis.numel(),
                                           _ptr(phis), int(igli), float(wind), lp, _ptr(out), self._stream()), "sosgpu_trphi")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().sosgpu_destroy(self._h)            # waits for the streams this context's work was queued on
            self._h = C.c_void_p()
            self._rsurf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_scratch():
    """Return the scratch buffers the library keeps from destroyed contexts (streamed solver; at most 8 GiB) to the device
    (sosgpu_trim)."""
    capi.check(capi.lib().sosgpu_trim(), "sosgpu_trim")


class ContextTable:
    """Device-resident table of wavelength contexts (sosgpu_ctx_table) for solve_spectrum: ONE kernel launch then covers the
    bins of all these wavelengths.  The contexts must agree in N, iborm_max and IMAT_SURF; the table stays valid while they
    live (rebuild it after closing one or changing its surface matrices)."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        if not self.ctxs:
            raise ValueError("ContextTable needs at least one context")
        self.device = self.ctxs[0].device
        L = capi.lib()
        n = len(self.ctxs)
        self.table = torch.empty(n * int(L.sosgpu_ctx_table_entry_bytes()), dtype=torch.uint8, device=self.device)
        hs = (C.c_void_p * n)(*[cx._h for cx in self.ctxs])
        capi.check(L.sosgpu_ctx_table(hs, n, _ptr(self.table), self.ctxs[0]._stream()), "sosgpu_ctx_table")


def concat_bins(bins_list):
    """Concatenate per-wavelength bin dicts (upload_bins / make_profiles) for solve_spectrum: the level axis is padded to the
    largest lp.  Returns (bins, ctx_of_bin int32 device tensor, seg int32 device tensor of len(bins_list) + 1)."""
    d = bins_list[0]["prof"].device
    lp = max(b["lp"] for b in bins_list)
    zo = [b["jout"] is not None for b in bins_list]
    if any(zo) != all(zo):
        raise ValueError("either every wavelength or none has an output level (zout)")
    prof = []
    for b in bins_list:
        p = b["prof"]
        if b["lp"] < lp:
            p = torch.nn.functional.pad(p, (0, lp - b["lp"]))
        prof.append(p)
    cat = lambda k: torch.cat([b[k] for b in bins_list])
    counts = [b["nb"] for b in bins_list]
    out = dict(nb=int(sum(counts)), lp=lp, perm=None, nt=cat("nt"), iborm=cat("iborm"), prof=torch.cat(prof).contiguous(),
               jout=cat("jout") if all(zo) else None, zz=cat("zz") if all(zo) else None)
    if all(b.get("scal") is not None for b in bins_list):
        out["scal"] = cat("scal")
    cob = _dev_i32(np.repeat(np.arange(len(bins_list), dtype=np.int32), counts), d)
    seg = _dev_i32(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), d)
    return out, cob, seg


def solve_spectrum(table, bins, ctx_of_bin, seg, aik, out=None, order="cost"):
    """The bin loops of MANY wavelengths (one SOS_PROC call each in the reference, binding/run_sos.py:640) as ONE launch of the
    fused solver plus one segmented SOS_AGGREGATE: bin b runs with the operators of table.ctxs[ctx_of_bin[b]], segment g of
    `seg` is wavelength g.  bins / ctx_of_bin / seg from concat_bins; aik[nb] device tensor in the same order.
    Returns (rec[nwavelengths][smax+1][3][W], scal[nwavelengths][10+N]) like SosContext.aggregate; no synchronisation."""
    cx = table.ctxs[0]
    if out is None:
        out = cx.alloc_outputs(bins["nb"], zero=False)
    # order="cost": the workgroups take the bins by decreasing total optical depth (levels x scattering orders go with it), so
    # that the long bins start first and the launch does not end on them; the bins stay where they are.  None: as given.
    # (Measured, scripts/spectrum_stream_bench.py / spectrum_bench.py: +4 % on real level grids, 160 x 25 bins; -5 % for the
    #  LDS-resident kernel at NT = 30, where neighbouring bins of one wavelength share their operators in L2 -- not applied there.)
    ord_t = None
    if isinstance(order, str) and order == "cost" and bins["lp"] <= 64:
        order = None
    if isinstance(order, str) and order == "cost":
        htot = bins["prof"][:, 0, :].gather(1, bins["nt"].long().clamp(min=0, max=bins["lp"] - 1)[:, None])[:, 0]
        ord_t = torch.argsort(htot, descending=True, stable=True).to(torch.int32)
    elif order is not None:
        ord_t = _dev_i32(order, cx.device)
    capi.check(capi.lib().sosgpu_os_solve_multi(cx._h, _ptr(table.table), _ptr(ctx_of_bin), _ptr(ord_t), bins["nb"], bins["lp"],
                                                _ptr(bins["nt"]), _ptr(bins["iborm"]), _ptr(bins["prof"]),
                                                _ptr(bins["jout"]), _ptr(bins["zz"]), _ptr(out["rec"]), _ptr(out["norders"]),
                                                _ptr(out["iglast"]), _ptr(out["flux"]), cx._stream()), "sosgpu_os_solve_multi")
    return cx.aggregate(out, aik, seg=seg, scal=bins.get("scal"))


def solve_many(items, n_streams=16):
    """Hyperspectral shape of the work (BASELINE config 5): MANY wavelengths with FEW CKD bins each.  One wavelength = one
    SosContext (its own source operators); its few bins occupy a fraction of the chip (one workgroup per bin, 512 resident),
    so the wavelengths are issued round-robin on `n_streams` HIP streams and overlap on the device.
    items: list of (ctx, bins, aik) with bins from ctx.upload_bins / make_profiles and aik a DEVICE tensor (a host array would
    be copied after the solve on the same stream, which blocks the host and serialises everything).
    Returns [(rec, scal)] of ctx.aggregate per wavelength, in order; the caller synchronises (torch.cuda.synchronize()).
    The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): export GPU_MAX_HW_QUEUES=16 before
    the first GPU call -- measured on 128 wavelengths x 32 bins: 18.6 k bins/s on one stream, 47.9 k (4 queues), 76.4 k (8),
    125 k (16) (scripts/spectrum_bench.py, profiles/r02_spectrum.txt)."""
    if not items:
        return []
    dev = items[0][0].device
    cur = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(n_streams, len(items))))]
    for st in streams:
        st.wait_stream(cur)
    res = []
    for i, (cx, bins, aik) in enumerate(items):
        with torch.cuda.stream(streams[i % len(streams)]):
            out = cx.solve(bins, cx.alloc_outputs(bins["nb"], zero=False))
            res.append(cx.aggregate(out, aik, scal=bins.get("scal")))
    for st in streams:
        cur.wait_stream(st)
    return res
