// csrc/mie.hip -- Mie theory for homogeneous spheres on gfx950.
//
// Replaces SOS_MIE (src/SOS_MIE.F:205-715) and SOS_FPHASE_MIE (:801-945): for every size parameter alpha of the grid the
// reference writes one record {alpha, Qext, Qsca (REAL*4), g (double), Imie, Qmie, Umie (REAL*4) at the 2 NBMU + 1 scattering
// angles} to the MIE cache file; here the records stay in HBM and feed the size-distribution integral (aerosols.py, SOS_GRANU).
// One workgroup per size parameter -- they are independent.  The Mie coefficients a_n, b_n come from serial recurrences
// (Ricatti-Bessel functions upwards, logarithmic derivatives downwards, exactly the reference's statement order): thread 0
// runs them; then one thread per scattering angle sums the amplitude functions S1, S2 over n with the pi_n / tau_n
// recurrences.  FP64 throughout, results rounded to REAL*4 like the file records.  The 11 arrays of 2 alpha + 24 terms sit in LDS
// up to alpha ~ 840 and in an HBM scratch beyond (up to the reference's CTE_MIE_DIM = 10000 terms).
#include "sos_common.h"
#include "kernels.h"
#include <algorithm>

#pragma clang fp contract(off)

// rec[a][4 + 3 W] floats: alpha, qext, qsca, 0, Imie[W], Qmie[W], Umie[W] (W = 2 nbmu + 1, index j + nbmu, j = -nbmu..nbmu);
// g[a] double.  xmu[W] = RMU(-nbmu:nbmu).
// GLOBAL = false: the 11 coefficient arrays (nmax entries each) live in LDS -- size parameters up to ~840;
// GLOBAL = true : they live in a per-workgroup slot of an HBM scratch (the WMO dust-like component needs alpha = 4000,
//                 SOS.h:122).  The recurrences carry their previous terms in registers, so the serial chain never waits
//                 for its own stores; the arithmetic and its order are the same in both forms.
// Workgroup b handles the size parameters first + b, first + b + gridDim.x, ... below `last`.
template <bool GLOBAL>
__global__ void k_mie(int nbmu, const double *__restrict__ xmu, double rn, double in, const double *__restrict__ alphas,
                      int first, int last, int nmax, double *__restrict__ gscratch, float *__restrict__ rec,
                      double *__restrict__ gout, int32_t *__restrict__ err)
{
    extern __shared__ double lds[];
    double *sm = GLOBAL ? gscratch + (size_t)blockIdx.x * 11 * nmax : lds;
    // arrays with Fortran lower bound -1: element i at [i + 1]
    double *cna = sm, *sna = cna + nmax, *rgna = sna + nmax, *igna = rgna + nmax;
    double *rdna = igna + nmax, *rdnb = rdna + nmax, *idnb = rdnb + nmax;
    double *ra = idnb + nmax, *ia = ra + nmax, *rb = ia + nmax, *ib = rb + nmax;
    __shared__ int s_n2;
    __shared__ double s_q[3];
    const int t = threadIdx.x;
    const int W = 2 * nbmu + 1;
    for (int a = first + blockIdx.x; a < last; a += gridDim.x) {
        const double alpha = alphas[a];
        if (t == 0) {
            int n1 = (int)(alpha + alpha + 20), n2 = (int)(alpha + alpha + 5);
            if (n1 + 3 > nmax) { atomicOr(err, 1); s_n2 = -1; }
            else {
#define A(x, i) x[(i) + 1]
                // (every entry read below is written first: no clearing pass)
                double c2 = -sin(alpha), c1 = cos(alpha), rg = 0., ig = -1.;       // CNA(-1), CNA(0), RGNA(0), IGNA(0)
                A(cna, -1) = c2; A(cna, 0) = c1;
                A(rgna, -1) = 0.; A(rgna, 0) = rg; A(igna, -1) = 0.; A(igna, 0) = ig;
                for (int i = 1; i <= n2; i++) {                                  // SOS_MIE.F:455-470
                    const double x = rg, z = i / alpha, y = ig;
                    const double w = ((z - x) * (z - x) + (y * y));
                    rg = (z - x) / w - z;
                    ig = y / w;
                    const double c0 = (2 * i - 1.) * c1 / alpha - c2;
                    A(rgna, i) = rg; A(igna, i) = ig; A(cna, i) = c0;
                    c2 = c1; c1 = c0;
                    if (!(c0 < 1.e304)) { n2 = i; n1 = i + 15; break; }
                }
                const double rbeta = rn * alpha, ibeta = in * alpha;
                double x1 = rbeta * rbeta + ibeta * ibeta;
                double x2 = rbeta / x1, x3 = ibeta / x1;
                double nb_r = 0., nb_i = 0., na_r = 0., s_up = 0., s_at = 1.;     // RDNB/IDNB/RDNA(i+1), SNA(i+1), SNA(i)
                A(rdna, n1) = 0.; A(rdnb, n1) = 0.; A(idnb, n1) = 0.; A(sna, n1) = 0.; A(sna, n1 - 1) = 1.;
                for (int i = n1 - 1; i >= 0; i--) {                              // :482-503
                    double z = nb_r + (i + 1.) * x2, w = nb_i - (i + 1.) * x3;
                    const double x4 = z * z + w * w;
                    nb_r = (i + 1.) * x2 - z / x4;
                    nb_i = -(i + 1.) * x3 + w / x4;
                    z = (i + 1.) / alpha;
                    na_r = z - 1. / (na_r + z);
                    double s_lo = (2. * i + 1.) * s_at / alpha - s_up;
                    A(rdnb, i) = nb_r; A(idnb, i) = nb_i; A(rdna, i) = na_r; A(sna, i - 1) = s_lo;
                    s_up = s_at;
                    if (s_lo > 1.e304) {
                        const int test = i - 1;
                        const double xx = s_lo;
                        for (int j = test; j <= n2; j++) A(sna, j) = A(sna, j) / xx;
                        s_lo = A(sna, test);
                        s_up = A(sna, test + 1);
                    }
                    s_at = s_lo;
                }
                double q = A(sna, 0) / sin(alpha);
                for (int i = 0; i <= n2; i++) A(sna, i) = A(sna, i) / q;
                double un = 1;
                for (int i = 1; i <= n2; i++) {                                  // :509-555
                    const double x1 = A(sna, i), x2 = A(cna, i), x3 = A(rdnb, i), x4 = A(idnb, i), x5 = A(rdna, i);
                    const double x6 = A(rgna, i), x7 = A(igna, i);
                    double y1 = x3 - rn * x5, y2 = x4 - in * x5, y3 = x3 - rn * x6 + in * x7, y4 = x4 - rn * x7 - in * x6;
                    const double y5 = rn * x3 - in * x4 - x5, y6 = in * x3 + rn * x4, y7 = rn * x3 - in * x4 - x6,
                                 y8 = in * x3 + rn * x4 - x7;
                    const double z4 = y2 * y3 - y1 * y4, z3 = y1 * y3 + y2 * y4, z5 = x1 * x1 + x2 * x2, z6 = y3 * y3 + y4 * y4;
                    const double z7 = y5 * y7 + y6 * y8, z8 = y6 * y7 - y5 * y8, z9 = y7 * y7 + y8 * y8;
                    q = (i + i + 1.) / i / (i + 1.) * un;
                    if (x2 > 1.e300) { y1 = 0.; y2 = 0.; y3 = 0.; y4 = 0.; }
                    else {
                        y1 = x1 * (x1 * z3 + x2 * z4) / z5 / z6;
                        y2 = x1 * (x1 * z4 - x2 * z3) / z5 / z6;
                        y3 = x1 * (x1 * z7 + x2 * z8) / z5 / z9;
                        y4 = x1 * (x1 * z8 - x2 * z7) / z5 / z9;
                    }
                    ra[i] = y2 * q; ib[i] = y3 * q;
                    q = -q;
                    rb[i] = y4 * q; ia[i] = y1 * q;
                    un = -un;
                }
                ra[0] = 0.; ia[0] = 0.; rb[0] = 0.; ib[0] = 0.;
                ra[n2 + 1] = 0.; ia[n2 + 1] = 0.; rb[n2 + 1] = 0.; ib[n2 + 1] = 0.;
                double qext = 0., qsca = 0., g = 0.;
                int j = -1;
                double x = ra[1], y = ia[1], z = rb[1], tt0 = ib[1];
                for (int n = 1; n <= n2; n++) {                                  // :572-588
                    const int m = n + 1;
                    const double xx = ra[m], yy = ia[m], zz = rb[m], tt = ib[m];
                    const double a2 = (n + 1.);
                    qext = qext + n * a2 * j * (y - tt0);
                    qsca = qsca + n * n * a2 * a2 / (n + a2) * (x * x + y * y + z * z + tt0 * tt0);
                    j = -j;
                    g = g - a2 * n / (a2 + n) * (n * (a2 + 1.) * (a2 + 1.) / (2. * n + 3.) * (y * yy + x * xx + tt0 * tt + z * zz) + y * tt0 + x * z);
                    x = xx; y = yy; z = zz; tt0 = tt;
                }
                const double w6 = 2. / alpha / alpha;
                qext = w6 * qext; qsca = w6 * qsca;
                g = 4. * g / qsca / alpha / alpha;
                s_q[0] = qext; s_q[1] = qsca; s_q[2] = g;
                s_n2 = n2;
#undef A
            }
        }
        if (GLOBAL) __threadfence_block();
        __syncthreads();
        const int n2 = s_n2;
        if (n2 >= 0) {
            float *r = rec + (size_t)a * (4 + 3 * W);
            if (t == 0) { r[0] = (float)alpha; r[1] = (float)s_q[0]; r[2] = (float)s_q[1]; r[3] = 0.f; gout[a] = s_q[2]; }
            const double coef = 2. / s_q[1] / (alpha * alpha);
            for (int jj = t; jj < W; jj += blockDim.x) {                         // SOS_FPHASE_MIE :873-900
                const double x = -xmu[jj];
                double pim = 0., piv = 1., tau = x, res1 = 0., res2 = 0., ims1 = 0., ims2 = 0.;
                for (int n = 1; n <= n2; n++) {
                    const double ai = ia[n], bi = ib[n], ar = ra[n], br = rb[n];
                    res1 = res1 - ai * piv - bi * tau;
                    res2 = res2 + ai * tau + bi * piv;
                    ims1 = ims1 + ar * piv + br * tau;
                    ims2 = ims2 - ar * tau - br * piv;
                    const double pip = ((2. * n + 1.) * x * piv - (n + 1.) * pim) / n;
                    pim = piv; piv = pip;
                    tau = (n + 1.) * x * piv - (n + 2.) * pim;
                }
                const double y1 = res1 * res1 + ims1 * ims1, y2 = res2 * res2 + ims2 * ims2;
                const double y3 = 2. * res2 * res1, y4 = 2. * ims2 * ims1;
                r[4 + jj] = (float)(coef * (y1 + y2));
                r[4 + W + jj] = (float)(coef * (y2 - y1));
                r[4 + 2 * W + jj] = (float)(coef * (y3 + y4));
            }
        }
        __syncthreads();                                                         // the arrays and s_q are reused by the next alpha
    }
}

// alphas must ascend (alpha_grid does).  Size parameters whose arrays fit LDS run in the LDS form; the rest in the scratch
// form with MIE_SLOTS workgroups.  d_scratch: mie_scratch_doubles(alpha_max, number of scratch-form size parameters) doubles.
#define MIE_LDS_BYTES (150 * 1024)
#define MIE_SLOTS 2048
static int mie_nmax(double alpha) { return (int)(2 * alpha + 24); }
size_t mie_scratch_doubles(double alpha_max, int count)
{
    // (whether a scratch is needed is the caller's split of the grid at its LDS limit, sosgpu_mie: a grid ending between that
    //  limit, alpha = 850, and the 860 the LDS could hold still runs its tail in the scratch form)
    return (size_t)std::min(std::max(count, 0), MIE_SLOTS) * 11 * mie_nmax(alpha_max);
}

int launch_mie(int nalpha, int nbmu, const double *d_xmu, double rn, double in, const double *d_alphas, int n_lds, double alpha_lds,
               double alpha_max, double *d_scratch, float *d_rec, double *d_g, int32_t *d_err, hipStream_t st)
{
    // [0, n_lds): size parameters <= alpha_lds (LDS form); [n_lds, nalpha): scratch form
    if (n_lds > 0) {
        const int nmax = mie_nmax(alpha_lds);
        const size_t lds = (size_t)11 * nmax * sizeof(double);
        if (lds > MIE_LDS_BYTES) return -3;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mie<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        k_mie<false><<<n_lds, 128, lds, st>>>(nbmu, d_xmu, rn, in, d_alphas, 0, n_lds, nmax, nullptr, d_rec, d_g, d_err);
    }
    if (n_lds < nalpha) {
        if (!d_scratch) return -3;
        const int nmax = mie_nmax(alpha_max);
        const int wg = std::min(nalpha - n_lds, MIE_SLOTS);
        k_mie<true><<<wg, 128, 0, st>>>(nbmu, d_xmu, rn, in, d_alphas, n_lds, nalpha, nmax, d_scratch, d_rec, d_g, d_err);
    }
    return 0;
}

// SOS_GRANU (src/SOS_AEROSOLS.F:4392-4820) on the device: the integral of the Mie records over the size distribution, so that
// the records (2 MB per wavelength at 40 Mie angles) never travel to the host.  The reference reads the MIE file record by
// record and ACCUMULATES in file order (:4600-4618); here pass 1 forms the per-record weights in parallel and pass 2 adds them
// up in record order, one thread per output (3 W phase-function entries + the three scalars), so the sums are the Fortran
// loop's sums term for term.  igranu 1: log-normal (v1 modal radius, v2 ln-std); 2: Junge (v1 = r0, v2 slope, v3 = rmax).
//   out[0..2] = KMAT1 / SOMME_NR, KMAT2 / SOMME_NR, SOMME_NR;  out[3 + c W + j] = P11 | P12 | P33 (normalised by KMAT2)
// work[3 na + 1]: x1 qext | qsca x1 | nr pr, and the number of records used (the loop's exit, :4520 / :4585).
__device__ __forceinline__ void granu_block(int na, int nbmu, const float *__restrict__ rec, int igranu, double v1, double v2, double v3,
                                            double wa, double alphaf, double *__restrict__ work, double *__restrict__ out)
{
    __shared__ int s_nuse;
    const int t = threadIdx.x, W = 2 * nbmu + 1, RS = 4 + 3 * W;
    const double pi = 3.141592653589793;
    if (t == 0) s_nuse = na;
    __syncthreads();
    auto step_of = [](float a) {           // the REAL*4 step ladder of SOS_GRANU (:4523-4527), same literals as SOS_MIE's
        float pas = 0.0001f;
        if (a > 0.10f) pas = 0.001f;
        if (a > 1.00f) pas = 0.01f;
        if (a > 10.f) pas = 0.05f;
        if (a > 30.f) pas = 0.10f;
        if (a > 100.f) pas = 1.00f;
        return pas;
    };
    double *wq = work, *ws = work + na, *wn = work + 2 * (size_t)na;
    for (int i = t; i < na; i += blockDim.x) {
        const float af = rec[(size_t)i * RS];
        const double a64 = (double)af;
        const double r = a64 * wa / 2. / pi;
        const float pas = step_of(af);
        const double pas_prev = (double)(i ? step_of(rec[(size_t)(i - 1) * RS]) : 0.0001f);
        bool stop = a64 >= (alphaf - pas_prev);                                      // IF (ALPHA.GE.(ALPHAF-PAS)) GOTO 40
        double nr;
        if (igranu == 1) {
            const double b = log(r / v1) / v2;
            nr = exp(-b * b / 2.) / (r * v2 * sqrt(2 * pi));
        } else {
            if (r > v3) stop = true;                                                 // IF (R.GT.RMAX) GOTO 40
            nr = (r <= v1) ? pow(v1, -v2) : pow(r, -v2);
        }
        if (stop) atomicMin(&s_nuse, i);
        const double pr = wa * (double)pas / 2. / pi;
        const double x1 = nr * pr * pi * (r * r);
        wq[i] = x1 * (double)rec[(size_t)i * RS + 1];
        ws[i] = (double)rec[(size_t)i * RS + 2] * x1;
        wn[i] = nr * pr;
    }
    __syncthreads();
    const int nuse = s_nuse;
    __shared__ double s_k[3];
    // pass 2: one thread per output (3 W phase-function sums + the three scalar sums) adds its column in record order.  The
    // loads do not depend on the running sum: UB records are requested together (coalesced over the threads: the 3 W entries
    // of a record are contiguous), then added one after the other -- the serial chain is the additions only.
    constexpr int UB = 24;
    const int NO = 3 * W + 3;
    for (int o = t; o < NO; o += blockDim.x) {
        double acc = 0.;
        if (o < 3) {
            const double *src = work + (size_t)o * na;
            int i = 0;
            for (; i + UB <= nuse; i += UB) {
                double v[UB];
#pragma unroll
                for (int k = 0; k < UB; k++) v[k] = src[i + k];
#pragma unroll
                for (int k = 0; k < UB; k++) acc = acc + v[k];
            }
            for (; i < nuse; i++) acc = acc + src[i];
            s_k[o] = acc;
        } else {
            const float *col = rec + 4 + (o - 3);
            int i = 0;
            for (; i + UB <= nuse; i += UB) {
                float v[UB];
                double wv[UB];
#pragma unroll
                for (int k = 0; k < UB; k++) { v[k] = col[(size_t)(i + k) * RS]; wv[k] = ws[i + k]; }
#pragma unroll
                for (int k = 0; k < UB; k++) acc = acc + (double)v[k] * wv[k];
            }
            for (; i < nuse; i++) acc = acc + (double)col[(size_t)i * RS] * ws[i];
            out[o] = acc;                                                            // normalised below
        }
    }
    __syncthreads();
    const double kmat1 = s_k[0], kmat2 = s_k[1], somme = s_k[2];
    for (int o = t; o < 3 * W + 3; o += blockDim.x) {
        if (o == 0) out[0] = kmat1 / somme;
        else if (o == 1) out[1] = kmat2 / somme;
        else if (o == 2) out[2] = somme;
        else out[o] = out[o] / kmat2;
    }
    if (t == 0) work[3 * (size_t)na] = (double)nuse;
}

// Several size integrals in one launch, one workgroup each (the wavelengths of a spectrum: run_sos.sos_spectrum asks for the
// integrals of a batch of wavelengths ahead of their host preparation).  The jobs travel as the kernel argument (32 x 64 bytes).
struct GranuJobs { sosgpu_granu_job j[SOSGPU_GRANU_JOBS_PER_LAUNCH]; };

__global__ __launch_bounds__(256) void k_granu_batch(const GranuJobs jobs, int nbmu, double *__restrict__ work, size_t work_stride,
                                                     double *__restrict__ out)
{
    const sosgpu_granu_job &jb = jobs.j[blockIdx.x];
    granu_block(jb.nalpha, nbmu, jb.d_rec, jb.igranu, jb.v1, jb.v2, jb.v3, jb.wa, jb.alphaf, work + blockIdx.x * work_stride,
                out + (size_t)blockIdx.x * (3 + 3 * (2 * nbmu + 1)));
}

void launch_granu(int na, int nbmu, const float *d_rec, int igranu, double v1, double v2, double v3, double wa, double alphaf,
                  double *d_work, double *d_out, hipStream_t st)
{
    void launch_granu_batch(int, int, const sosgpu_granu_job *, double *, size_t, double *, hipStream_t);
    sosgpu_granu_job jb = {d_rec, na, igranu, v1, v2, v3, wa, alphaf};       // (one kernel for both entry points: the same sums)
    launch_granu_batch(1, nbmu, &jb, d_work, (size_t)3 * na + 1, d_out, st);
}

void launch_granu_batch(int count, int nbmu, const sosgpu_granu_job *jobs, double *d_work, size_t work_stride, double *d_out,
                        hipStream_t st)
{
    const size_t nout = (size_t)3 + 3 * (2 * nbmu + 1);
    for (int c0 = 0; c0 < count; c0 += SOSGPU_GRANU_JOBS_PER_LAUNCH) {
        const int n = count - c0 < SOSGPU_GRANU_JOBS_PER_LAUNCH ? count - c0 : SOSGPU_GRANU_JOBS_PER_LAUNCH;
        GranuJobs jb = {};
        for (int k = 0; k < n; k++) jb.j[k] = jobs[c0 + k];
        k_granu_batch<<<n, 256, 0, st>>>(jb, nbmu, d_work + (size_t)c0 * work_stride, work_stride, d_out + (size_t)c0 * nout);
    }
}
