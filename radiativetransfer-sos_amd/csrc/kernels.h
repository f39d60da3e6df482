// csrc/kernels.h -- host-side launchers of the gfx950 kernels (internal to libsosgpu.so).
#pragma once
#include "../../include/sosgpu.h"
#include "sos_common.h"

void launch_noyaux(const SosDev &cx, hipStream_t st);
void launch_noyaux_fetch(const SosDev &cx, int s, double *d_out, hipStream_t st);

// Fused successive-orders solver.  Returns 0, SOSGPU_E_UNSUPPORTED when (N, max NT) has no variant, or -2 with the HIP
// error code in *hip_err.
int launch_sos_os(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err);
// waves / row-tiles-per-wave / column tiles / LDS bytes the solver would use (for planning and tests); <0 if unsupported.
int sos_os_variant(int n, int nt_max, int *nw, int *rtw, int *ct, size_t *lds_bytes, int *big);
// Streamed-field solver (sos_stream.hip) for level grids beyond the LDS-resident variants: same contract as launch_sos_os.
// bn.scratch holds bn.scr_stride >= sos_stream_scratch_doubles(n, lpb) doubles per bin, lpb a multiple of 32.
int launch_sos_stream(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err);
size_t sos_stream_scratch_doubles(int n, int lpb);
// order-parallel form (bn.spec_k): the Fourier stop tests of orders [s0, s1) of every bin, after the order tasks of a round
int launch_sos_stream_replay(const SosDev &cx, const SosBins &bn, int s0, int s1, hipStream_t st, int *hip_err);
int sos_stream_threads(int n);
// the same two solvers with a per-bin context (bn.ctxs, bn.ctx_of_bin; sos_os_multi.hip, sos_stream_multi.hip): `cx` is any
// context of the table -- it selects the variant (N, surface-matrix flag), which the caller guarantees equal over the table
int launch_sos_os_multi(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err);
int launch_sos_stream_multi(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err);

void launch_aggregate(const SosDev &cx, int nseg, const int32_t *d_seg, const double *d_aik,
                      const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                      const double *d_tdifmug, double *d_out_rec, double *d_out_scal, hipStream_t st, int nb_single,
                      double *d_partial, int max_chunks);

void launch_glitter(int n, const double *d_mu, double sig, int os_nb, int os_ns, int os_nm, const double *d_fcoef,
                    int32_t *d_il, double *d_e, float *d_rsurf, hipStream_t st);
// pieces of the surface-matrix chains (glitter.hip / land.hip): azimuth quadrature of SOS_GSF (model 0: Cox-Munk, par = sigma^2)
// or SOS_GSF_MAIGNAN (model 1, par = C exp(-NDVI)); SOS_MAT_REFLEXION + SOS_MISE_FORMAT with COEF = 1/sigma^2 or 1
void launch_gsf(int model, int n, const double *d_mu, double par, int os_nm, int32_t *d_il, double *d_e, hipStream_t st);
void launch_mat_reflexion(int n, const double *d_mu, double coef, int os_nb, int os_ns, int os_nm, const double *d_fcoef,
                          const int32_t *d_il, const double *d_e, float *d_rsurf, hipStream_t st);
// land surfaces (-SURF.Type 3, 4, 5, 7): Roujean BRDF, + Rondeaux-Herman / Breon / Maignan BPDF
void launch_land(int isurf, int n, const double *d_mu, double k0, double k1, double k2, double coef_c, int os_nb, int os_ns, int os_nm, const double *d_fcoef, double *d_e_nn, int32_t *d_il_nn,
                 double *d_e, int32_t *d_il, float *d_tmp, float *d_rsurf, int32_t *d_err, hipStream_t st);
struct LandTerms {            // direct surface terms of the land models in SOS_TRPHI (SOS_PREPA_OS.F:479-497 flags)
    int iroujean, irondeaux, ibreon, imaignan;
    double k0, k1, k2, coef_c;
};
void launch_trphi(const SosDev &cx, int nf, const double *d_rec, double tau, double tauout, int nphi,
                  const double *d_phi, int igli, double sigma2, double ind_surf, const LandTerms &land, double *d_out,
                  hipStream_t st);

#define SOS_PROF_NBLEV_MAX 64     // levels of the absorption profile held in LDS (CTE_ABS_NBLEV = 50 in the reference)
// Per-bin profile discretisation (profile.hip).  *_ng: the no-gas profile of the wavelength (host-computed, device copy).
struct ProfileArgs {
    int nb, lp, nblev, absprofil, smax, nt_ng;
    double tr, hr, ta, ha, a_tronc, piz, piztr, zout;
    const double *altabs, *tabs;                 // [nblev], [nb][nblev] or null
    const double *z_ng, *h_ng, *pca_ng, *pcm_ng;  // [nt_ng+1]
    double *prof, *zprof, *zz, *scal;
    int32_t *nt, *iborm, *jout;
};
void launch_profile(const ProfileArgs &a, hipStream_t st);
// the no-gas profile of the wavelength into d_ng = z | h | pca | pcm, `ng` doubles each (nt + 1 <= ng used)
void launch_profile_nogas(double tr, double hr, double ta, double ha, int nt, double t_first, double t_layer, double *d_ng, int ng,
                          hipStream_t st);

// SOS_ABSPROFILE for nb bins: ik[nb][8] 1-based term per gas, xk[8][nterm][nlev-1], ro[8][nlev-1] -> tabs[nb][nlev]
void launch_absprofile(int nb, int nlev, int nterm, const int32_t *d_ik, const double *d_xk, const double *d_ro, double *d_tabs,
                       hipStream_t st);

// Mie records of a size-parameter grid (mie.hip): rec[nalpha][4 + 3 (2 nbmu + 1)] floats, g[nalpha]; returns 0, -2 (HIP) or -3
// (alpha_max beyond the LDS-resident coefficient arrays)
size_t mie_scratch_doubles(double alpha_max, int count);
// SOS_GRANU on the device records: d_work[3 na + 1], d_out[3 + 3 (2 nbmu + 1)] (mie.hip)
void launch_granu(int na, int nbmu, const float *d_rec, int igranu, double v1, double v2, double v3, double wa, double alphaf,
                  double *d_work, double *d_out, hipStream_t st);
// ... for `count` jobs, one workgroup each: d_work[count][work_stride], d_out[count][3 + 3 (2 nbmu + 1)]
void launch_granu_batch(int count, int nbmu, const sosgpu_granu_job *jobs, double *d_work, size_t work_stride, double *d_out,
                        hipStream_t st);
int launch_mie(int nalpha, int nbmu, const double *d_xmu, double rn, double in, const double *d_alphas, int n_lds, double alpha_lds,
               double alpha_max, double *d_scratch, float *d_rec, double *d_g, int32_t *d_err, hipStream_t st);
