// csrc/glitter.hip -- Cox-Munk rough-sea reflection matrices on gfx950.
//
// Replaces SOS_GLITTER (reference src/SOS_GLITTER.F:229-371), which chains four routines through
// temporary files:
//   SOS_GSF          (SOS_GLITTER.F:451-711)  wave-facet azimuth quadrature, one Fourier series per angle pair
//   SOS_MAT_FRESNEL  (SOS_SURFACE.F:1235-1603) Legendre expansion of the Fresnel matrix -- O(N*OS_NS) work with a
//                    4(E15.8) TEXT round trip: done on the host in api.hip (host_mat_fresnel), bit for bit
//   SOS_MAT_REFLEXION + SOS_NOYAUX_FRESNEL (SOS_SURFACE.F:1708-1973, 2029-2227)  per-pair Fourier matrices
//   SOS_MISE_FORMAT  (SOS_SURFACE.F:2307-2443) re-layout into per-order N x N REAL*4 matrices (81 re-reads of
//                    a 12 MB file in the reference; here it is just the store index)
//
// Kernel k_gsf: one wavefront per angle pair (I1 >= I2).  The 1025 samples of the facet function G(phi)
// live in LDS; for every Fourier order the recursive-halving trapezoid (<= 10 levels, 1..512 cosine
// terms per level) is summed across the 64 lanes with shuffle reductions; the data-dependent stops
// (relative 1e-4 per level, 1e-3 series closure -> IL) are wave-uniform.  No HBM traffic besides the
// (IL, E) result: the kernel is FP64 transcendental (cos/exp) bound.
// Kernel k_mat_reflexion: one workgroup per pair; thread k builds the Fresnel kernels of Fourier index k
// (sequential Legendre recurrence + sums in the reference's order), then thread s forms the 17 sums over
// k of order s (reference order, un-fused) and stores the REAL*4 results directly in FICSURF record order.
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)

#define PH_NU 1024
#define PH_NQ 10
#define PH_TEST 10000

// MODEL 0: Cox-Munk facet function, SOS_CALCG (SOS_GLITTER.F:779-781), par = sigma^2
// MODEL 1: Maignan BPDF, SOS_CALCG_MAIGNAN (SOS_SURFACE_BPDF.F:1606-1641), par = C exp(-NDVI); cs12 then holds 1/C1 + 1/C2
template <int MODEL>
__device__ __forceinline__ double calcg(double cs12, double c12, double s12, double par, double phi)
{
    if (MODEL == 0) {
        const double costetad = -c12 + s12 * cos(phi);
        const double x = (1 - costetad) / cs12;
        return x * x * exp(-(x - 1) / par);
    }
    const double cos2i = c12 - s12 * cos(phi);
    double tan2i = (1 - cos2i) / (1 + cos2i);
    if (tan2i < 0.) tan2i = 0.;
    return par * exp(-sqrt(tan2i)) / cs12;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// il[pair], e[pair][os_nm+1] (zero beyond IL); pairs ordered (I1 = 1..N, I2 = 1..I1).  The same quadrature serves
// SOS_GSF (MODEL 0) and SOS_GSF_MAIGNAN (MODEL 1, SOS_SURFACE_BPDF.F:1305-1600): only the function differs.
template <int MODEL>
__global__ __launch_bounds__(64) void k_gsf(const double *__restrict__ mu, double sig, int os_nm,
                                           int32_t *__restrict__ il_out, double *__restrict__ e_out)
{
    __shared__ double u[PH_NU + 1];
    const int pair = blockIdx.x, lane = threadIdx.x;
    int i1 = 0;
    while ((i1 + 1) * (i1 + 2) / 2 <= pair) i1++;
    const int i2 = pair - i1 * (i1 + 1) / 2;     // 0-based, i2 <= i1
    const double pi = acos(-1.0);
    const double c1 = mu[i1], s1 = sqrt(1 - c1 * c1), c2 = mu[i2], s2 = sqrt(1 - c2 * c2);
    const double c12 = c1 * c2, s12 = s1 * s2;
    double cs12 = (c1 + c2);
    cs12 = .5 * cs12 * cs12;
    if (MODEL == 1) cs12 = 1. / c1 + 1. / c2;
    const double gmax = calcg<MODEL>(cs12, c12, s12, sig, 0.0);
    double gmin = calcg<MODEL>(cs12, c12, s12, sig, pi);
    double phib, q;
    bool full = (PH_TEST * gmin >= gmax);
    if (full) {                                   // SOS_GLITTER.F:568-578
        phib = pi;
        q = pi / PH_NU;
    } else {                                      // bisection, SOS_GLITTER.F:586-620 (every lane, identical)
        double phi1 = 0, phi2 = pi;
        for (int it = 0; it < 200; it++) {
            phib = .5 * (phi1 + phi2);
            const double x = PH_TEST * calcg<MODEL>(cs12, c12, s12, sig, phib);
            if (fabs(x - gmax) < (double).01f * gmax) break;
            if (x <= gmax) phi2 = phib; else phi1 = phib;
        }
        q = phib / PH_NU;
    }
    for (int i = lane; i <= PH_NU; i += 64) u[i] = (i == 0) ? gmax : calcg<MODEL>(cs12, c12, s12, sig, q * i);
    __syncthreads();
    if (!full) gmin = u[PH_NU];
    double *e = e_out + (size_t)pair * (os_nm + 1);
    for (int s = lane; s <= os_nm; s += 64) e[s] = 0.;
    __syncthreads();
    double t1 = 0.;
    int il = os_nm;
    for (int is = 0; is <= os_nm; is++) {         // SOS_GLITTER.F:644-678
        double z = .5 * (gmax + gmin * cos(is * phib));
        int ia = 1;
        for (int lev = 1; lev <= PH_NQ; lev++) {
            ia = 2 * ia;
            const int ip = PH_NU / ia;
            double y = 0.;
            for (int t = lane; t < ia / 2; t += 64) {     // odd J = 2t+1
                const int k = ip * (2 * t + 1);
                y = y + u[k] * cos((is * k) * q);
            }
            y = wave_sum(y);
            y = 2 * y / ia;
            const double xt = fabs(z - y) / z;
            if (xt < (double).0001f) break;
            z = .5 * (y + z);
        }
        const double es = phib * z / pi;
        if (lane == 0) e[is] = es;
        if (is == 0) { t1 = es; continue; }
        t1 = t1 + 2 * es;
        if (!(fabs(t1 - gmax) / gmax > (double).001f)) { il = is; break; }
    }
    if (lane == 0) il_out[pair] = il;
}

// Fresnel kernels of SOS_NOYAUX_FRESNEL for Fourier index `is` at the angle pair (r1, r2); results
// kout[X*2 + (K-1)], X = 0 BP, 1 GR, 2 GT, 3 ARR, 4 ART, 5 ATT; K = 1,2 (J = 3-K), SOS_SURFACE.F:2198-2223.
__device__ void noyaux_fresnel_one(int is, double r1, double r2, int os_ns, const double *__restrict__ al,
                                   const double *__restrict__ be, const double *__restrict__ ga,
                                   const double *__restrict__ ze, double *kout)
{
    const double r[2] = {r1, r2};
    double pl[2], plm[2], rl[2], rlm[2], tl[2], tlm[2];
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0.;
    // one Legendre degree: add the l-term with values (p, rr, t) at both angles
    auto add = [&](int l, const double *p, const double *rr, const double *t) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int j = 1 - k;
            acc[0 + k] = acc[0 + k] + be[l] * p[j] * p[k];
            acc[2 + k] = acc[2 + k] + ga[l] * p[j] * rr[k];
            acc[4 + k] = acc[4 + k] + ga[l] * p[j] * t[k];
            acc[10 + k] = acc[10 + k] + al[l] * t[j] * t[k] + ze[l] * rr[j] * rr[k];
            acc[6 + k] = acc[6 + k] + ze[l] * t[j] * t[k] + al[l] * rr[j] * rr[k];
            acc[8 + k] = acc[8 + k] + al[l] * rr[k] * t[j] + ze[l] * rr[j] * t[k];
        }
    };
    int lstart;
    if (is == 0) {                                // SOS_SURFACE.F:2107-2118
        const double x26 = 2. * sqrt(6.0);
        double p0[2] = {1., 1.}, zero[2] = {0., 0.};
        add(0, p0, zero, zero);
        double p1[2] = {r[0], r[1]};
        if (os_ns >= 1) add(1, p1, zero, zero);
        for (int j = 0; j < 2; j++) {
            const double c = r[j];
            pl[j] = (3 * c * c - 1) * 0.5; plm[j] = c;
            rl[j] = 3 * (1 - c * c) / x26; rlm[j] = 0.;
            tl[j] = 0.; tlm[j] = 0.;
        }
        lstart = 2;
    } else if (is == 1) {                         // SOS_SURFACE.F:2124-2134
        const double rac3 = sqrt(3.0);
        double p1[2], zero[2] = {0., 0.};
        for (int j = 0; j < 2; j++) {
            const double c = r[j], x = 1 - c * c;
            p1[j] = sqrt(x * 0.5);
            pl[j] = c * p1[j] * rac3; plm[j] = p1[j];
            rl[j] = -c * sqrt(x) * 0.5; rlm[j] = 0.;
            tl[j] = -sqrt(x) * 0.5; tlm[j] = 0.;
        }
        add(1, p1, zero, zero);
        lstart = 2;
    } else {                                      // SOS_SURFACE.F:2139-2159
        double a = 1;
        for (int i = 1; i <= is; i++) { const double x = i; a = a * sqrt((i + is) / x) * 0.5; }
        const double b = a * sqrt(is / (is + 1.0)) * sqrt((is - 1.0) / (is + 2.));
        for (int j = 0; j < 2; j++) {
            const double c = r[j], xx = 1 - c * c;
            double yy = is * 0.5;
            double x = pow(xx, yy);
            pl[j] = a * x;
            yy = yy - 1;
            x = pow(xx, yy);
            rl[j] = b * (1 + c * c) * x;
            tl[j] = 2 * b * c * x;
            plm[j] = 0.; rlm[j] = 0.; tlm[j] = 0.;
        }
        lstart = is;
    }
    if (lstart <= os_ns) add(lstart, pl, rl, tl);
    for (int l = lstart; l <= os_ns - 1; l++) {   // SOS_SURFACE.F:2167-2189
        const double a = (2 * l + 1.) / sqrt((l + is + 1.0) * (l - is + 1.));
        const double b = sqrt((double)((l + is) * (l - is))) / (2. * l + 1.);
        const double d = (l + 1.) * (2 * l + 1.) / sqrt((l + 3.0) * (l - 1.) * (l + is + 1.) * (l - is + 1.));
        const double e = sqrt((l + 2.0) * (l - 2.) * (l + is) * (l - is)) / (l * (2. * l + 1.));
        const double f = (double)((2.f * (float)is) / ((float)l * ((float)l + 1.f)));   // REAL*4, SOS_SURFACE.F:2176
        double pn[2], rn[2], tn[2];
        for (int j = 0; j < 2; j++) {
            const double c = r[j];
            pn[j] = a * (c * pl[j] - b * plm[j]);
            rn[j] = d * (c * rl[j] - f * tl[j] - e * rlm[j]);
            tn[j] = d * (c * tl[j] - f * rl[j] - e * tlm[j]);
            plm[j] = pl[j]; rlm[j] = rl[j]; tlm[j] = tl[j];
        }
        for (int j = 0; j < 2; j++) { pl[j] = pn[j]; rl[j] = rn[j]; tl[j] = tn[j]; }
        add(l + 1, pl, rl, tl);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) kout[i] = acc[i];
}

// rsurf[((s*9 + ab)*N + (J-1))*N + (I-1)] = P_ab(I,J)
__global__ void k_mat_reflexion(int n, const double *__restrict__ mu, double coef, int os_nb, int os_ns, int os_nm,
                                const double *__restrict__ fcoef /* [4][os_ns+1] */, const int32_t *__restrict__ il_in,
                                const double *__restrict__ e_in, float *__restrict__ rsurf)
{
    extern __shared__ double sm[];
    double *g = sm;                                // [os_nm+1]
    double *kern = g + (os_nm + 1);                // [os_ns+1][12]
    const int pair = blockIdx.x, t = threadIdx.x;
    int i1 = 0;
    while ((i1 + 1) * (i1 + 2) / 2 <= pair) i1++;
    const int i2 = pair - i1 * (i1 + 1) / 2;
    const int I = i1 + 1, J = i2 + 1;
    const int lim = il_in[pair];
    for (int k = t; k <= os_nm; k += blockDim.x) g[k] = (k <= lim) ? e_in[(size_t)pair * (os_nm + 1) + k] : 0.;
    const double *al = fcoef, *be = fcoef + (os_ns + 1), *ga = fcoef + 2 * (os_ns + 1), *ze = fcoef + 3 * (os_ns + 1);
    for (int k = t; k <= os_ns; k += blockDim.x) noyaux_fresnel_one(k, mu[i1], mu[i2], os_ns, al, be, ga, ze, kern + 12 * k);
    __syncthreads();
#define KX(X, k, c) kern[12 * (k) + 2 * (X) + (c)-1]
    for (int is = t; is <= os_nb; is += blockDim.x) {   // SOS_SURFACE.F:1864-1933
        double x = coef * g[is] / 4., y;
        double r111 = x * KX(0, 0, 1), r121 = x * KX(1, 0, 1), r122 = x * KX(1, 0, 2), r131 = 0., r132 = 0., r231 = 0., r232 = 0.;
        double r211 = x * KX(1, 0, 2), r212 = x * KX(1, 0, 1), r221 = x * KX(3, 0, 2), r222 = x * KX(3, 0, 1);
        double r311 = 0., r312 = 0., r321 = 0., r322 = 0., r331 = x * KX(5, 0, 2), r332 = x * KX(5, 0, 1);
        int im = 1;
        for (int k = 1; k <= os_ns; k++) {
            const int a1 = k + is, a2 = abs(k - is);
            im = -im;
            if ((a1 > lim) && (a2 > lim)) continue;
            x = coef * im * (g[a1] + g[a2]) / 4.;
            y = coef * im * (g[a2] - g[a1]) / 4.;
            r111 = r111 + KX(0, k, 1) * x;
            r121 = r121 + KX(1, k, 1) * x; r122 = r122 + KX(1, k, 2) * x;
            r131 = r131 + KX(2, k, 1) * y; r132 = r132 + KX(2, k, 2) * y;
            r211 = r211 + KX(1, k, 2) * x; r212 = r212 + KX(1, k, 1) * x;
            r221 = r221 + KX(3, k, 2) * x; r222 = r222 + KX(3, k, 1) * x;
            r231 = r231 + KX(4, k, 2) * y; r232 = r232 + KX(4, k, 1) * y;
            r311 = r311 + KX(2, k, 2) * y; r312 = r312 + KX(2, k, 1) * y;
            r321 = r321 + KX(4, k, 1) * y; r322 = r322 + KX(4, k, 2) * y;
            r331 = r331 + KX(5, k, 2) * x; r332 = r332 + KX(5, k, 1) * x;
        }
        // M(IS,1) -> P(I,J), M(IS,2) -> P(J,I); for I == J the second store wins (SOS_SURFACE.F:2378-2395)
        const double m1[9] = {r111, r121, r131, r211, r221, r231, -r311, -r321, -r331};
        const double m2[9] = {r111, r122, r132, r212, r222, r232, -r312, -r322, -r332};
        for (int ab = 0; ab < 9; ab++) {
            float *o = rsurf + ((size_t)is * 9 + ab) * n * n;
            o[(size_t)(J - 1) * n + (I - 1)] = (float)m1[ab];
            o[(size_t)(I - 1) * n + (J - 1)] = (float)m2[ab];
        }
    }
#undef KX
}

void launch_gsf(int model, int n, const double *d_mu, double par, int os_nm, int32_t *d_il, double *d_e, hipStream_t st)
{
    const int npairs = n * (n + 1) / 2;
    if (model == 0) k_gsf<0><<<npairs, 64, 0, st>>>(d_mu, par, os_nm, d_il, d_e);
    else k_gsf<1><<<npairs, 64, 0, st>>>(d_mu, par, os_nm, d_il, d_e);
}

void launch_mat_reflexion(int n, const double *d_mu, double coef, int os_nb, int os_ns, int os_nm, const double *d_fcoef,
                          const int32_t *d_il, const double *d_e, float *d_rsurf, hipStream_t st)
{
    const size_t sh = ((size_t)(os_nm + 1) + 12 * (size_t)(os_ns + 1)) * sizeof(double);
    k_mat_reflexion<<<n * (n + 1) / 2, 128, sh, st>>>(n, d_mu, coef, os_nb, os_ns, os_nm, d_fcoef, d_il, d_e, d_rsurf);
}

void launch_glitter(int n, const double *d_mu, double sig, int os_nb, int os_ns, int os_nm, const double *d_fcoef,
                    int32_t *d_il, double *d_e, float *d_rsurf, hipStream_t st)
{
    launch_gsf(0, n, d_mu, sig, os_nm, d_il, d_e, st);
    launch_mat_reflexion(n, d_mu, 1. / sig, os_nb, os_ns, os_nm, d_fcoef, d_il, d_e, d_rsurf, st);
}
