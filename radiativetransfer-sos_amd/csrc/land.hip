// csrc/land.hip -- land-surface reflection matrices on gfx950.
//
// Replaces, without their temporary files, the reference routines behind -SURF.Type 3..7 (SOS_SURFACE.F:640-990):
//   SOS_ROUJEAN            src/SOS_ROUJEAN.F:212    = SOS_FSF_ROUJEAN (:417) + SOS_CALC_F_ROUJEAN (:891) + SOS_MISE_FORMAT_RJ (:1102)
//   SOS_SURFACE_BPDF       src/SOS_SURFACE_BPDF.F:219 with its model-specific azimuth analyses
//       SOS_GSF_RONDEAUX_BREON (:463), SOS_GSF_MAIGNAN (:1305, kernel k_gsf<1> of glitter.hip) + SOS_CALCG_MAIGNAN (:1606)
//       (the Nadal model, -SURF.Type 6, is refused by the reference's own SOS_PROC -- "The Nadal's BPDF model is not
//       supported" -- and is not built),
//     followed by SOS_MAT_FRESNEL / SOS_MAT_REFLEXION(1.0, ...) / SOS_MISE_FORMAT (glitter.hip, shared with the sea surface)
//   SOS_BPDF_AJOUT_BRDF    src/SOS_SURFACE.F:2503   BPDF + Roujean BRDF, REAL*4 element by element
//
// k_fsf<MODEL>: one wavefront per ordered angle pair (I1, I2) of the N x N pairs.  The 1025 azimuth samples of the model
// function sit in registers (16 per lane + one); for every Fourier order the rectangle sum E(IS) (SOS_ROUJEAN.F:589-596)
// and the maximum relative error B1 of the recombined series (:601-623) are wave reductions, and the reference's stop rule
// (B1 <= 1e-3, or B1 growing) is wave-uniform.  FP64 transcendental bound (one cos per sample and order), no HBM traffic
// besides the result.
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)
#include "land_models.h"

#define PH_NU 1024

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// Roujean BRDF (p = K0, K1, K2); keeps E(IS) of every order it computed (the reference stores all of them, :640-643).
// e_out[pair][os_nb+1], pair = (I1-1)*N + (I2-1); err[0] |= 1 when the function goes negative (IER = -1, :548).
__global__ __launch_bounds__(64) void k_fsf(int n, const double *__restrict__ mu, int os_nb, double p0, double p1, double p2,
                                           int32_t *__restrict__ il_out, double *__restrict__ e_out, int32_t *__restrict__ err)
{
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int i1 = pair / n, i2 = pair % n;
    const double pi = acos(-1.0);
    const double c1 = mu[i1], s1 = sqrt(1 - c1 * c1), c2 = mu[i2], s2 = sqrt(1 - c2 * c2);
    const double q = pi / PH_NU;
    constexpr int NS = PH_NU / 64 + 1;                 // 16 samples per lane, lane 0 also holds sample 1024
    double u[NS], t1[NS];
    bool neg = false;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int i = (k < NS - 1) ? lane + 64 * k : PH_NU;
        const double phi = q * i;
        u[k] = calc_f_roujean(p0, p1, p2, c1, s1, c2, s2, pi - phi);
        if (k == NS - 1 && lane != 0) u[k] = 0.;
        if (u[k] < 0.) neg = true;
        t1[k] = 0.;
    }
    if (__any(neg)) { if (lane == 0) atomicOr(err, 1); }
    double *e = e_out + (size_t)pair * (os_nb + 1);
    for (int s = lane; s <= os_nb; s += 64) e[s] = 0.;
    double b1_prec = 1.e300;
    int il = os_nb;
    for (int is = 0; is <= os_nb; is++) {
        double cs[NS], y = 0.;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const int i = (k < NS - 1) ? lane + 64 * k : PH_NU;
            cs[k] = cos(is * (i * q));
            y = y + u[k] * cs[k];
        }
        y = wsum(y);
        const double es = y * q / pi;
        if (lane == 0) e[is] = es;
        double b1 = 0.;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            t1[k] = (is == 0) ? es : t1[k] + 2. * es * cs[k];
            if (k < NS - 1 || lane == 0) b1 = fmax(fabs((t1[k] - u[k]) / u[k]), b1);
        }
        b1 = wmax(b1);
        if (!(b1 > (double)0.001f)) { il = is; break; }            // CTE_SEUIL_SF_ROUJEAN (REAL*4 literal, SOS.h:339)
        if (!(b1 < b1_prec)) { il = is - 1; break; }
        b1_prec = b1;
    }
    if (lane == 0) il_out[pair] = il;
}

// Rondeaux-Herman (MODEL 0: E(0) = 1/(1/C1 + 1/C2)) and Breon (MODEL 1: E(0) = 1): azimuth-independent, IL = 0
// (SOS_SURFACE_BPDF.F:560-575).  Pairs (I1 >= I2) as SOS_MAT_REFLEXION reads them.
__global__ void k_gsf_const(int n, int model, const double *__restrict__ mu, int os_nm, int32_t *__restrict__ il_out,
                            double *__restrict__ e_out)
{
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n * (n + 1) / 2) return;
    int i1 = 0;
    while ((i1 + 1) * (i1 + 2) / 2 <= pair) i1++;
    const int i2 = pair - i1 * (i1 + 1) / 2;
    double *e = e_out + (size_t)pair * (os_nm + 1);
    for (int s = 0; s <= os_nm; s++) e[s] = 0.;
    e[0] = model == 0 ? 1. / (1. / mu[i1] + 1. / mu[i2]) : 1.;
    il_out[pair] = 0;
}

// SOS_MISE_FORMAT_RJ (SOS_ROUJEAN.F:1102-1224): P11(I,J) = REAL(E_(I,J)(IS)), every other element zero
__global__ void k_roujean_format(int n, int os_nb, const double *__restrict__ e_nn, float *__restrict__ rsurf)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)(os_nb + 1) * 9 * n * n;
    if (idx >= tot) return;
    const int i = (int)(idx % n), j = (int)((idx / n) % n), ab = (int)((idx / ((size_t)n * n)) % 9), s = (int)(idx / ((size_t)9 * n * n));
    rsurf[idx] = (ab == 0) ? (float)e_nn[((size_t)i * n + j) * (os_nb + 1) + s] : 0.f;
}

// SOS_BPDF_AJOUT_BRDF (SOS_SURFACE.F:2503-2670)
__global__ void k_add_f32(size_t cnt, const float *__restrict__ a, float *__restrict__ io)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < cnt) io[idx] = io[idx] + a[idx];
}

// isurf 3: Roujean; 4: + Rondeaux-Herman; 5: + Breon; 7: + Maignan
void launch_land(int isurf, int n, const double *d_mu, double k0, double k1, double k2, double coef_c, int os_nb, int os_ns, int os_nm, const double *d_fcoef, double *d_e_nn, int32_t *d_il_nn,
                 double *d_e, int32_t *d_il, float *d_tmp, float *d_rsurf, int32_t *d_err, hipStream_t st)
{
    const int npairs = n * (n + 1) / 2;
    const size_t cnt = (size_t)(os_nb + 1) * 9 * n * n;
    // Roujean BRDF (every land model carries it)
    k_fsf<<<n * n, 64, 0, st>>>(n, d_mu, os_nb, k0, k1, k2, d_il_nn, d_e_nn, d_err);
    float *brdf = isurf == 3 ? d_rsurf : d_tmp;
    k_roujean_format<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(n, os_nb, d_e_nn, brdf);
    if (isurf == 3) return;
    if (isurf == 4 || isurf == 5) k_gsf_const<<<(npairs + 63) / 64, 64, 0, st>>>(n, isurf == 4 ? 0 : 1, d_mu, os_nm, d_il, d_e);
    else launch_gsf(1, n, d_mu, coef_c, os_nm, d_il, d_e, st);
    launch_mat_reflexion(n, d_mu, 1.0, os_nb, os_ns, os_nm, d_fcoef, d_il, d_e, d_rsurf, st);
    k_add_f32<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(cnt, brdf, d_rsurf);
}
