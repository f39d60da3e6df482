/* include/sosgpu.h -- C ABI of libsosgpu.so, the MI355X (gfx950) drop-in for the SOS-ABS hot path.
 *
 * Plain C: pointers + sizes, int status (0 ok, negative = error, see sosgpu_strerror).  No torch
 * types, no Fortran hidden lengths, no files.  All `d_` pointers are DEVICE (HBM) addresses owned by
 * the caller (the Python host allocates them with torch); `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Calls are asynchronous on `stream` unless stated.
 *
 * Streams (the ordering rule).  No entry point synchronises the device, and none touches the null stream on its own:
 *   - entry points taking `stream` queue ALL their device work on it and -- unless documented "synchronous" -- return
 *     without waiting; their d_ inputs must have been produced on `stream` (or be complete) and must stay allocated
 *     until the queued work has run;
 *   - host-synchronous entry points without a stream argument (sosgpu_create, sosgpu_set_surface_matrices,
 *     sosgpu_noyaux_fetch, sosgpu_os_flops) move their data on a private non-blocking stream of the calling host thread
 *     (created at the highest stream priority: the runtime maps streams onto a few hardware queues, in order per queue,
 *     and at normal priority these short copies would wait behind the kernels of whatever caller stream shares the queue)
 *     and wait for that stream only; the last two first wait for the streams THIS context's work was queued on.  They
 *     write only memory the library owns (never a caller's buffer, which the caller's other queued work may still be
 *     using), and what they wrote is complete on return, so work queued afterwards on any stream sees it;
 *   - a context may be used from one host thread at a time; different contexts may be driven from different host
 *     threads on different streams concurrently (run_sos.sos_proc_many), and no call waits for another thread's work;
 *   - sosgpu_destroy waits for the streams the context's own work was queued on, nothing else.
 * Device memory of contexts and of temporaries is recycled through a process-wide pool (hipFree would synchronise the
 * device); sosgpu_trim() returns it.
 *
 * Each entry point replaces one routine of the reference per-wavelength pipeline that
 * binding/run_sos.py reaches through sos.sos_proc (binding/run_sos.py:640, SOS_PROC.F:415):
 *
 *   sosgpu_profile     <- SOS_PROFILE + SOS_DISC  src/SOS_PROFIL.F:224,1210 (+ the PROFIL read-back and rescale of
 *                                                 SOS, src/SOS.F:511-550; all bins of a wavelength at once)
 *   sosgpu_noyaux      <- SOS_NOYAUX              src/SOS_OS.F:1857   (phase-matrix Fourier kernels,
 *                                                                     hoisted out of the bin loop)
 *   sosgpu_os_solve    <- SOS_OS (+ leaves)       src/SOS_OS.F:303    (one call = a batch of CKD bins,
 *                                                                     i.e. the loop SOS_PROC.F:3459-3594)
 *   sosgpu_aggregate   <- SOS_AGGREGATE           src/SOS_AGGREGATE.F:172
 *   sosgpu_glitter     <- SOS_GLITTER (+SOS_GSF, SOS_MAT_FRESNEL, SOS_MAT_REFLEXION, SOS_MISE_FORMAT)
 *                                                 src/SOS_GLITTER.F:229, src/SOS_SURFACE.F:1235,1708,2307
 *   sosgpu_trphi       <- SOS_TRPHI               src/SOS_TRPHI.F:749
 *
 * Index conventions (identical to oracle/sos_oracle.h): N = NBMU positive directions, mu[0..N-1] =
 * RMU(1..N) descending; direction jj in -N..N lives at offset jj+N of width W = 2N+1 (slot jj=0 is
 * unused and written as 0); Fourier records rec[s][c][W], c = 0:I 1:Q 2:U; level 0 = TOA.
 */
#ifndef SOSGPU_H
#define SOSGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SOSGPU_OK            0
#define SOSGPU_E_ARG        -1   /* bad argument / inconsistent sizes (reference IER=-1) */
#define SOSGPU_E_HIP        -2   /* HIP runtime error (sosgpu_last_hip_error) */
#define SOSGPU_E_UNSUPPORTED -3  /* size outside the compiled kernel variants */
#define SOSGPU_E_NODEVICE   -4   /* no gfx950 device visible */
#define SOSGPU_E_RCCL       -5   /* librccl missing or an RCCL call failed */

/* Per-wavelength description (everything SOS_OS receives that does not depend on the CKD bin). */
typedef struct sosgpu_wave {
    int32_t n;          /* NBMU: Gauss + sun (+ user) positive directions, <= 80 (SOS.h:471) */
    int32_t os_nb;      /* max Legendre order of the phase-matrix expansion, <= 200 (SOS.h:480) */
    int32_t n0;         /* 1-based index of the solar direction in mu[] (N0 of SOS_OS); mus = mu[n0-1] */
    int32_t imat_surf;  /* 1: BRDF/BPDF matrices given in d_rsurf (SOS_OS.F:912-925) */
    int32_t ifresnel;   /* 1: flat-sea Fresnel reflection (SOS_OS.F:817,1010,1225) */
    int32_t ipolar;     /* 0: no polarisation (SOS_OS.F:689-699, 928-941), 1: normal */
    int32_t igmax;      /* max scattering order (CTE_DEFAULT_IGMAX = 100) */
    int32_t reserved;
    double  ro;         /* Lambertian albedo */
    double  ind_surf;   /* refractive index (Fresnel) */
    double  ron;        /* molecular depolarisation factor (CTE_MDF) */
} sosgpu_wave;

/* Opaque per-wavelength context living on one device. */
typedef struct sosgpu_ctx sosgpu_ctx;

const char *sosgpu_strerror(int code);
int  sosgpu_last_hip_error(void);
/* Number of visible gfx950 devices, or negative error. */
int  sosgpu_device_count(void);
const char *sosgpu_version(void);

/* Create / destroy a context on `device`.  Host arrays are copied: mu[n], ga[n] (Gauss weights, 0 for
 * the sun/user angles), alpha/beta/gamma/zeta[os_nb+1].  iborm_max = highest Fourier order any bin of
 * this wavelength may need (OS_NB, or 2 for a purely molecular atmosphere, SOS.F:549-550). */
int  sosgpu_create(sosgpu_ctx **out, int device, const sosgpu_wave *wv, const double *mu, const double *ga,
                   const double *alpha, const double *beta, const double *gamma, const double *zeta,
                   int iborm_max);
int  sosgpu_destroy(sosgpu_ctx *cx);

/* Surface reflection matrices for imat_surf=1: REAL*4, reference FICSURF record order
 * d_rsurf[s][ab][(J-1)*N+(I-1)] = R_ab(I,J), s = 0..iborm_max (SOS_OS.F:916-925).  Device pointer; the call
 * packs them into the context as FP64 ground-reflection operators in matrix-core fragment order (weights, 2/mu and the
 * Lambertian part folded in).  Host-synchronous: d_rsurf must be complete when the call is made and may be released when it
 * returns (see "Streams" above; no device-wide synchronisation).  Call it after sosgpu_create and before sosgpu_os_solve. */
int  sosgpu_set_surface_matrices(sosgpu_ctx *cx, const float *d_rsurf);
/* Stream-ordered form of the same: the packing is queued on `stream` and nothing is waited for (d_rsurf complete or being
 * produced on `stream`; keep it allocated until the stream has passed this point).  Solves queued later on the same stream
 * see the operators; solves on OTHER streams must be ordered after it by the caller (event / synchronise). */
int  sosgpu_set_surface_matrices_async(sosgpu_ctx *cx, const float *d_rsurf, void *stream);

/* Replaces SOS_NOYAUX (SOS_OS.F:1857-2158) for every Fourier order 0..iborm_max at once; must be called
 * once per context before sosgpu_os_solve.  Fills the context's packed source operators. */
int  sosgpu_noyaux(sosgpu_ctx *cx, void *stream);
/* Debug/parity accessor: copies the six kernels of order `is` to host as the reference lays them
 * out, X[(j+N)*W + (k+N)] = X(J,K), in the order BP,GR,GT,ARR,ART,ATT, then XPL,XRL,XTL (W each).
 * Synchronous.  out must hold 6*W*W + 3*W doubles. */
int  sosgpu_noyaux_fetch(sosgpu_ctx *cx, int is, double *out);

/* Replaces the per-bin SOS_OS calls of the CKD loop (SOS_PROC.F:3459-3594 -> SOS.F:554 -> SOS_OS.F:303).
 *  nb          number of bins in this batch
 *  lp          padded level count (row stride of d_prof), >= max(nt)+1
 *  d_nt[nb]    NT of each bin (int32)
 *  d_iborm[nb] IBORM of each bin (int32; SOS.F:549-550), <= iborm_max
 *  d_prof[nb][3][lp]   H, XDEL, YDEL after the truncation rescale of SOS.F:523-543
 *  d_jout[nb]  output level pair for ZOUT != -1: levels jout-1 and jout bracket ZOUT (SOS_OS.F:1514-1520);
 *              0 = standard output (ZOUT = -1: TOA up, ground down).  May be NULL (= all 0).
 *  d_zz[nb]    interpolation weight ZZ (SOS_OS.F:1520); ignored when jout = 0.  May be NULL.
 * outputs
 *  d_rec[nb][iborm_max+1][3][W]  Fourier records; only orders 0..d_norders[b]-1 hold records on return (the reference's FICOS
 *                                file of a bin holds one record per order run, SOS_OS.F:1571-1575): a buffer zero-filled by the
 *                                caller is zero beyond them (the order-parallel launch form computes a few orders past the
 *                                stop and clears their rows again)
 *  d_norders[nb]                 number of Fourier orders run (int32); -1 = malformed bin (NT < 1, NT >= lp, IBORM out of
 *                                range: the reference's IER = -1), nothing else is written for it
 *  d_iglast[nb][iborm_max+1]     last scattering order computed per Fourier order (int32)
 *  d_flux[nb][2]                 EMOINS, EPLUS (SOS_OS.F:1447-1456)
 */
int  sosgpu_os_solve(sosgpu_ctx *cx, int nb, int lp, const int32_t *d_nt, const int32_t *d_iborm,
                     const double *d_prof, const int32_t *d_jout, const double *d_zz,
                     double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux, void *stream);

/* The scratch of the streamed solver (level grids beyond 64 levels) is kept by the library when a context is destroyed and
 * handed to the next context that needs one (at most 8 buffers and 8 GiB per process): a context per wavelength would
 * otherwise pay a 60-1000 MB hipMalloc per call.  sosgpu_trim() returns that memory to the device. */
int  sosgpu_trim(void);

/* Many wavelengths in ONE launch (hyperspectral runs, BASELINE config 5: each wavelength has only 5-100 CKD bins, a fraction
 * of the 512 workgroups the chip hosts).  The reference runs SOS_PROC once per wavelength (binding/run_sos.py:640); here the
 * bin loops SOS_PROC.F:3459-3594 of nctx wavelengths are concatenated and every bin carries the index of its wavelength.
 *   sosgpu_ctx_table   copies the device-side description of nctx contexts (sosgpu_ctx_table_entry_bytes() each) into the
 *                      caller's device buffer d_table, ordered on `stream` (nothing is waited for: the copy comes from a pinned
 *                      block ctxs[0] keeps until it is destroyed).  All contexts must live on one device and agree in
 *                      N, iborm_max and IMAT_SURF (E_ARG otherwise).  The table refers to the contexts' operator tables: it
 *                      stays valid until one of them is destroyed or has sosgpu_set_surface_matrices called again -- a
 *                      destroyed context's memory is recycled at once, so wait for the launches that use the table first.
 *   sosgpu_os_solve_multi   sosgpu_os_solve with d_ctx_of_bin[nb] (int32, 0..nctx-1): bin b is solved with the operators of
 *                      table entry d_ctx_of_bin[b].  `cx` is any context of the table (it provides the variant selection, the
 *                      streamed variant's scratch and the timing events).  d_order[nb] (int32, a permutation of 0..nb-1) or
 *                      NULL: workgroup i solves bin d_order[i] -- workgroups start in index order, so listing the costliest
 *                      bins first shortens the tail of the launch while the bins stay grouped by wavelength in memory.
 *                      All other arguments as sosgpu_os_solve; feed sosgpu_aggregate with one segment per wavelength. */
size_t sosgpu_ctx_table_entry_bytes(void);
int  sosgpu_ctx_table(sosgpu_ctx *const *ctxs, int nctx, void *d_table, void *stream);
int  sosgpu_os_solve_multi(sosgpu_ctx *cx, const void *d_table, const int32_t *d_ctx_of_bin, const int32_t *d_order, int nb, int lp,
                           const int32_t *d_nt, const int32_t *d_iborm, const double *d_prof, const int32_t *d_jout,
                           const double *d_zz, double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux,
                           void *stream);

/* Replaces SOS_AGGREGATE (SOS_AGGREGATE.F:372-488) for nseg independent wavelengths/bands at once:
 * segment g covers bins seg[g]..seg[g+1]-1 of d_rec; seg[0] = 0, seg[nseg] = nb.  One big band (nseg = 1,
 * nb > 128) is reduced in chunks of 64 bins (deterministic; the strict serial bin order of the reference is kept for
 * small bands and multi-band calls).  nb = 0 (a rank whose shard of the band is empty) is allowed with nseg = 1: the
 * outputs are the neutral element of the cross-rank reduce.
 *  d_scal[nb][4]     per-bin scalars: TDIFMUS, TTOT_TRONC, TTOT_VRAI, TAUOUT
 *  d_tdifmug[nb][N]  per-bin diffuse transmissions TDIFMUG(1..N) of the -SOS.Trans option (SOS.F:611-635), or NULL
 *  d_out_rec[nseg][iborm_max+1][3][W] = sum_b aik[b] * rec[b]   (orders a bin did not run count as zero records,
 *                    SOS_AGGREGATE.F:357-413; bins with norders < 0 are skipped)
 *  d_out_scal[nseg][SOSGPU_SCAL_BASE + N]:
 *      [0..2] sum aik*{TDIFMUS, EMOINS, EPLUS}     [3..5] sum aik*exp(-{TTOT_TRONC, TTOT_VRAI, TAUOUT})
 *      [6]    sum aik        [7] max norders       [8] -(min norders): > 0 when a bin of the segment failed
 *      [9]    0              [10..10+N) sum aik*TDIFMUG(j)     (SOS_AGGREGATE.F:452-459)
 *   Across GPUs elements 7 and 8 combine with MAX, all others (and d_out_rec) with SUM; the -ln of the three
 *   transmissions (SOS_AGGREGATE.F:467-488) is applied afterwards (sosgpu_reduce does all of this).
 */
#define SOSGPU_SCAL_BASE 10
int  sosgpu_aggregate(sosgpu_ctx *cx, int nb, int nseg, const int32_t *d_seg, const double *d_aik,
                      const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                      const double *d_tdifmug, double *d_out_rec, double *d_out_scal, void *stream);

/* Cross-GPU step of SOS_AGGREGATE for callers without torch.distributed (C / Fortran hosts): one RCCL
 * all-reduce (ncclDouble, ncclSum) over xGMI of the packed buffer d_buf[nseg][(iborm_max+1)*3*W + SOSGPU_SCAL_BASE + N]
 * (records followed by the scalar block, the layout sosgpu_pack writes) plus one 2-element MAX all-reduce per segment
 * for elements 7 and 8.  `comm` is an ncclComm_t (as void*) the caller created -- sosgpu_comm_* wrap ncclGetUniqueId /
 * ncclCommInitRank / ncclCommDestroy so that a host needs no RCCL headers.  librccl is resolved at the first call
 * (dlopen), so single-GPU users carry no RCCL dependency.  nranks = 1 is a no-op. */
#define SOSGPU_UNIQUE_ID_BYTES 128
int  sosgpu_comm_unique_id(char id[SOSGPU_UNIQUE_ID_BYTES]);
int  sosgpu_comm_init_rank(void **comm, int nranks, const char id[SOSGPU_UNIQUE_ID_BYTES], int rank);
int  sosgpu_comm_destroy(void *comm);
/* d_out_rec / d_out_scal of sosgpu_aggregate -> d_buf (device-to-device copies on `stream`) and back. */
int  sosgpu_pack(sosgpu_ctx *cx, int nseg, const double *d_out_rec, const double *d_out_scal, double *d_buf, void *stream);
int  sosgpu_unpack(sosgpu_ctx *cx, int nseg, const double *d_buf, double *d_out_rec, double *d_out_scal, void *stream);
int  sosgpu_reduce(sosgpu_ctx *cx, void *comm, int nseg, double *d_buf, void *stream);

/* Replaces SOS_GLITTER (SOS_GLITTER.F:229-371: SOS_GSF + SOS_MAT_FRESNEL + SOS_MAT_REFLEXION +
 * SOS_MISE_FORMAT, no temporary files).  Host inputs mu[n], chr[n] (Gauss weights), wind (m/s), ind (water
 * refractive index), orders OS_NB, OS_NS, OS_NM (>= OS_NB+OS_NS).  Device outputs:
 *  d_rsurf[os_nb+1][9][N][N]  REAL*4 matrices in GLITTER-file record order (feed to
 *                             sosgpu_set_surface_matrices with imat_surf = 1)
 *  d_il[N(N+1)/2]             series length IL of each angle pair (I1 = 1..N, I2 = 1..I1), SOS_GLITTER.F:676
 *  d_e[N(N+1)/2][os_nm+1]     Fourier coefficients E(0:IL) of the facet function, zero beyond IL
 * Synchronous with respect to the host arrays; kernels run on `stream`. */
int  sosgpu_glitter(int device, int n, const double *mu, const double *chr, double wind, double ind,
                    int os_nb, int os_ns, int os_nm, float *d_rsurf, int32_t *d_il, double *d_e, void *stream);
/* Host helper used by sosgpu_glitter, exposed for parity tests: SOS_MAT_FRESNEL (SOS_SURFACE.F:1235-1603)
 * including the 4(E15.8) text round trip; out[4][os_ns+1] = alpha, beta, gamma, zeta. */
int  sosgpu_mat_fresnel_host(int n, const double *mu, const double *chr, double ind, int os_ns, double *out);

/* Replaces SOS_TRPHI (SOS_TRPHI.F:749-1243) + SOS_POLAR (:1843) for nphi azimuths at once.
 *  nf           number of Fourier orders in d_rec (records beyond are ignored)
 *  d_rec[nf][3][W]  aggregated Fourier records (sosgpu_aggregate output, after the cross-GPU reduce)
 *  tau, tauout  total (truncated) optical depth and optical depth of the output level (TTOT_TRONC, TAUOUT)
 *  d_phi[nphi]  azimuths in radians
 *  igli         1: add the directly reflected Cox-Munk glint (needs wind); the flat-sea sun glint is added
 *               when the context has ifresnel = 1
 *  d_out[nphi][7][W]: XIT, XQT, XUT, ANGDIFF (deg), polarisation angle, rate (%), polarised radiance;
 *               slot jj = 0 is zero.
 *  land         NULL, or the land-surface model of -SURF.Type 3..7 whose directly reflected term is added
 *               (SOS_TRPHI.F:1047-1200): Roujean BRDF for every type, plus the BPDF of type 4..7. */
typedef struct sosgpu_land {
    int32_t isurf;            /* 3 Roujean, 4 + Rondeaux-Herman, 5 + Breon, 7 + Maignan (SOS_PREPA_OS.F:479-497); 6 (Nadal) is
                               * refused with SOSGPU_E_UNSUPPORTED, as the reference's SOS_PROC refuses it */
    int32_t reserved;
    double  k0, k1, k2;       /* Roujean coefficients */
    double  alpha, beta;      /* Nadal (kept for layout compatibility, unused) */
    double  coef_c;           /* Maignan C exp(-NDVI) */
} sosgpu_land;
int  sosgpu_trphi(sosgpu_ctx *cx, int nf, const double *d_rec, double tau, double tauout, int nphi,
                  const double *d_phi, int igli, double wind, const sosgpu_land *land, double *d_out, void *stream);

/* Replaces SOS_ROUJEAN (src/SOS_ROUJEAN.F:212), SOS_SURFACE_BPDF (src/SOS_SURFACE_BPDF.F:219) and SOS_BPDF_AJOUT_BRDF
 * (src/SOS_SURFACE.F:2503) for -SURF.Type 3..7, no temporary files: Fourier reflection matrices of the land surface,
 *  d_rsurf[os_nb+1][9][N][N]  REAL*4, reference surface-file record order (feed to sosgpu_set_surface_matrices).
 * Host inputs mu[n], chr[n]; ind = surface refractive index (BPDF types).  *ier_out (host) = 0, or -1 when the Roujean BRDF
 * goes negative for some geometry (the reference's IER = -1, SOS_ROUJEAN.F:548).  Synchronous. */
int  sosgpu_land_surface(int device, const sosgpu_land *land, int n, const double *mu, const double *chr, double ind,
                         int os_nb, int os_ns, int os_nm, float *d_rsurf, int32_t *ier_out, void *stream);

/* Scratch requirements (bytes) of the context on its device, for memory planning. */
size_t sosgpu_ctx_bytes(const sosgpu_ctx *cx);

/* Floating-point work of the last solve, computed on the host from d_nt/d_norders/d_iglast after a synchronise
 * (used by bench.py for the roofline).  flops_out[0] = SURVEY 8d count of the reference algorithm: every computed
 * scattering order >= 2 costs 2*(6N)^2*(NT+1) [+ 2*3*6N*(NT+1)*3 for the molecular operator, s <= 2] + 12*6N*NT.
 * flops_out[1] = the same steps in the form this library executes: two 3N x 3N half systems
 * 2*2*(3N)^2*(NT+1), the molecular operator in rank-4 form 2*2*4*3N*(NT+1) for s <= 2, and 10 flops per row and
 * layer of formal solution (6N*NT*10); padding to MFMA tiles is not counted. */
int  sosgpu_os_flops(sosgpu_ctx *cx, int nb, const int32_t *d_nt, const int32_t *d_norders,
                     const int32_t *d_iglast, double *flops_out);

/* Time (ms) of the last sosgpu_os_solve kernel, measured with HIP events on its own stream.
 * Synchronises that stream. */
int  sosgpu_last_solve_ms(sosgpu_ctx *cx, float *ms);

/* Per-bin atmospheric profiles on the device: SOS_PROFILE for IPROFIL = 1 (src/SOS_PROFIL.F:224-1170) with SOS_DISC
 * (:1210-1332), the PROFIL-file round trip (formats F10.5 / E15.8, SOS_PROFIL.F:1084,1150 -> SOS.F:515,692), the truncation
 * rescale and IBORM of SOS (src/SOS.F:521-550), TAUOUT / TTOT (SOS.F:567-589) and the output level of SOS_OS.F:1514-1520,
 * for nb CKD bins at once.  Replaces the per-bin calls `CALL SOS_PROFILE` (SOS_PROC.F:3518) + the read in `SOS`.
 *   tr, hr, ta, ha      Rayleigh / aerosol optical thickness and scale heights of the wavelength
 *   d_tabs[nb][nblev]   cumulative gas absorption optical depth of every bin on the altitude grid d_altabs[nblev]
 *                       (descending, ground last; CTE_ABS_NBLEV = 50 in the reference); NULL = no gas (ABSPROFIL = 7)
 *   a_tronc, piz, piztr truncation coefficient and single-scattering albedos of SOS.F:523-541;  zout: -1 = TOA/ground
 * Outputs (device): d_prof[nb][3][lp] (H, XDEL, YDEL as sosgpu_os_solve takes them), d_nt[nb] (-1: the profile needs
 * more than CTE_OS_NT = 600 levels or lp is too small -- the reference's IER = -1), d_iborm[nb], d_zprof[nb][lp],
 * d_jout[nb] / d_zz[nb] (NULL when zout = -1), d_scal[nb][4] = {0, TTOT_TRONC, TTOT_VRAI, TAUOUT} (the layout
 * sosgpu_aggregate takes).  The no-gas profile of the wavelength (SOS_PROFIL.F:349-489) is made by one wavefront queued on
 * `stream` in front of the bins' kernel, or taken from d_nogas (sosgpu_profile_nogas below); nothing is waited for (a second call on the same context from ANOTHER stream first
 * waits for the context's earlier work).  IPROFIL = 2 (aerosol layer between two altitudes) is not implemented
 * (the reference's branch reads an unassigned Hmol(0), its output is not reproducible). */
int  sosgpu_profile(sosgpu_ctx *cx, int nb, double tr, double hr, double ta, double ha, int absprofil,
                    int nblev, const double *d_altabs, const double *d_tabs,
                    double a_tronc, double piz, double piztr, double zout, int lp,
                    double *d_prof, int32_t *d_nt, int32_t *d_iborm, double *d_zprof,
                    int32_t *d_jout, double *d_zz, double *d_scal, const double *d_nogas, void *stream);

/* Head start for sosgpu_profile (optional).  The level placement of a wavelength's no-gas profile is a serial chain of about a
 * millisecond on one wavefront and needs (tr, hr, ta, ha) only: a driver can queue it here as soon as it knows them -- before it
 * has the gas tables, the surface or even the context of the wavelength -- and hand the block to sosgpu_profile as d_nogas
 * (NULL there: sosgpu_profile queues the same kernel itself, in front of the bins' kernel).
 *   d_nogas[4][SOSGPU_NOGAS_LEVELS] (DEVICE, caller-owned): altitude, optical depth, aerosol and molecular share of each level;
 *   must be complete, or queued on the same stream, when sosgpu_profile runs, and stay allocated until that work has run.
 * Asynchronous on `stream`.  SOSGPU_E_UNSUPPORTED: more than CTE_OS_NT levels (the reference's IER = -1). */
#define SOSGPU_NOGAS_LEVELS 608
int  sosgpu_profile_nogas(int device, double tr, double hr, double ta, double ha, double *d_nogas, void *stream);

/* Replaces the per-bin calls `CALL SOS_ABSPROFILE` of the CKD loop (SOS_PROC.F:3494; src/SOS_ABSPROFILE.F:184, core :325-371)
 * for nb bins at once.  The coefficient of a gas depends on the gas, the exponential term and the layer only, so the host
 * tabulates it once per wavelength (COEFF_ABS_CKD, src/SOS_SUB_TRS.F:171; absorption.layer_tables) and a bin is one term
 * index per gas:
 *   d_ik[nb][8]              1-based term index IK1..IK8 of each bin (gas order H2O, CO2, O3, N2O, CO, CH4, O2, NO2)
 *   d_xk[8][nterm][nlev-1]   k_i of (gas, term, layer), layer 0 = top layer;  d_ro[8][nlev-1] molecules/cm2 of the layer
 *   d_tabs[nb][nlev]         TAUABS: cumulative absorption optical depth per level, level 0 = TOA (feeds sosgpu_profile)
 * nlev = CTE_ABS_NBLEV = 50 in the reference. */
int  sosgpu_absprofile(int device, int nb, int nlev, int nterm, const int32_t *d_ik, const double *d_xk, const double *d_ro,
                       double *d_tabs, void *stream);

/* Replaces SOS_MIE + SOS_FPHASE_MIE (src/SOS_MIE.F:205, :801) for a whole grid of size parameters, no MIE cache file:
 *   xmu[2 nbmu + 1]  cosines RMU(-nbmu:nbmu) of the Mie angle set (host); rn, in: refractive index (in <= 0)
 *   alphas[nalpha]   size parameters, ascending (host; the reference's grid: steps 1e-4 ... 1 growing with alpha,
 *                    SOS_MIE.F:437-443)
 *   d_rec[nalpha][4 + 3 (2 nbmu + 1)]  REAL*4 records {alpha, Qext, Qsca, 0, Imie(-nbmu:nbmu), Qmie(..), Umie(..)}
 *   d_g[nalpha]      asymmetry factor (double, as the file keeps it)
 * The Mie coefficient arrays (2 alpha + 24 terms) live in LDS up to alpha = 850 and in a temporary HBM scratch beyond;
 * SOSGPU_E_UNSUPPORTED past the reference's own dimension (CTE_MIE_DIM = 10000 terms, SOS.h:96).  Synchronous. */
int  sosgpu_mie(int device, int nbmu, const double *xmu, double rn, double in, int nalpha, const double *alphas,
                float *d_rec, double *d_g, void *stream);

/* Replaces SOS_GRANU (src/SOS_AEROSOLS.F:4392-4820): the integral of Mie records over a size distribution, on the device -- the
 * records (2 MB per refractive index at 40 Mie angles) never travel to the host.  The reference reads its MIE file record by
 * record and accumulates in file order; the kernel adds in record order too (the sums of the Fortran loop term for term).
 *   d_rec[nalpha][4 + 3 (2 nbmu + 1)]  records as sosgpu_mie wrote them (the reference re-uses a MIE file for every
 *          wavelength with the same refractive index, angle set and size-parameter range: keep d_rec the same way)
 *   igranu 1: log-normal distribution, v1 = modal radius (microns), v2 = ln-standard deviation (v3 unused);
 *          2: Junge's law, v1 = r0, v2 = slope, v3 = rmax;   wa = wavelength (microns);  alphaf = upper limit of the grid
 *   out[3 + 3 (2 nbmu + 1)] (HOST): extinction and scattering cross sections per particle KMAT1, KMAT2, the number integral
 *          SOMME_NR, then P11, P12, P33 at the 2 nbmu + 1 angles (normalised as SOS_GRANU leaves them).  Synchronous. */
int  sosgpu_granu(int device, int nbmu, int nalpha, const float *d_rec, int igranu, double v1, double v2, double v3,
                  double wa, double alphaf, double *out, void *stream);

/* The same integral for `count` (records, distribution, wavelength) jobs at once, ASYNCHRONOUS on `stream` (the wavelengths of
 * a spectrum: run_sos.sos_spectrum queues the integrals of a batch of wavelengths ahead of their host preparation and fetches
 * all results with one copy).  One workgroup per job, SOSGPU_GRANU_JOBS_PER_LAUNCH jobs per launch; a job's sums are those of
 * sosgpu_granu bit for bit (the same code per workgroup).
 *   jobs[count] (HOST; read before the call returns)   one sosgpu_granu argument set each, all with the same nbmu
 *   d_out[count][3 + 3 (2 nbmu + 1)] (DEVICE)          the `out` block of sosgpu_granu per job
 *   d_work[count][work_stride] (DEVICE)                work_stride >= 3 max(nalpha) + 1 doubles; free after `stream` passed the call
 * Errors: SOSGPU_E_ARG for a malformed job (nothing is launched), SOSGPU_E_HIP. */
#define SOSGPU_GRANU_JOBS_PER_LAUNCH 32
typedef struct sosgpu_granu_job {
    const float *d_rec;                 /* records of sosgpu_mie (device) */
    int32_t nalpha, igranu;
    double v1, v2, v3, wa, alphaf;
} sosgpu_granu_job;
int  sosgpu_granu_batch(int device, int nbmu, int count, const sosgpu_granu_job *jobs, double *d_out, double *d_work,
                        size_t work_stride, void *stream);

/* Diagnostic hook: hand the context a device buffer [nb][8] of uint64 that builds compiled with
 * -DSOS_PROFILE_PHASES fill with per-phase cycle sums of the solver kernel (0 order-1 fill, 1 formal solution,
 * 2 contraction, 3 write-back, 4 stop tests, 5 ground boundary, 6 Fourier bookkeeping).  NULL disables.
 * The shipped build never writes it. */
int  sosgpu_debug_phase_buffer(sosgpu_ctx *cx, unsigned long long *d_phase);
/* Diagnostic accessor: device pointer and size (doubles) of the streamed solver's scratch of this context, and the offset of the
 * order-parallel form's I3 hand-over block [nb][iborm_max+1][threads] inside it after such a solve (0 otherwise). */
int  sosgpu_debug_scratch(sosgpu_ctx *cx, double **d_scratch, size_t *doubles, size_t *spec_i3_offset);

#ifdef __cplusplus
}
#endif
#endif
