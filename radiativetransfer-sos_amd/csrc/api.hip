// csrc/api.hip -- extern "C" surface of libsosgpu.so (see include/sosgpu.h for the contract and the
// reference routine each entry point replaces).
#include "../../include/sosgpu.h"
#include "kernels.h"
#include "sos_common.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

static thread_local int g_last_hip = 0;
#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t e_ = (x);                       \
        if (e_ != hipSuccess) {                    \
            g_last_hip = (int)e_;                  \
            return SOSGPU_E_HIP;                   \
        }                                          \
    } while (0)

struct sosgpu_ctx {
    int device;
    SosDev d;
    std::vector<void *> allocs;
    size_t bytes;
    hipEvent_t ev0, ev1;
    bool timed;
    hipStream_t last_stream;
    int nt_max_hint;
};

extern "C" const char *sosgpu_version(void) { return "sosgpu 0.1 (gfx950)"; }
extern "C" int sosgpu_last_hip_error(void) { return g_last_hip; }

extern "C" const char *sosgpu_strerror(int code)
{
    switch (code) {
    case SOSGPU_OK: return "ok";
    case SOSGPU_E_ARG: return "bad argument";
    case SOSGPU_E_HIP: return "HIP runtime error";
    case SOSGPU_E_UNSUPPORTED: return "problem size outside the compiled kernel variants";
    case SOSGPU_E_NODEVICE: return "no gfx950 device visible";
    default: return "unknown error";
    }
}

extern "C" int sosgpu_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_last_hip = (int)e; return SOSGPU_E_NODEVICE; }
    return n;
}

template <typename T>
static int dev_alloc(sosgpu_ctx *cx, T **p, size_t count)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, count * sizeof(T)));
    cx->allocs.push_back(q);
    cx->bytes += count * sizeof(T);
    *p = static_cast<T *>(q);
    return 0;
}

extern "C" int sosgpu_create(sosgpu_ctx **out, int device, const sosgpu_wave *wv, const double *mu, const double *ga,
                             const double *alpha, const double *beta, const double *gamma, const double *zeta,
                             int iborm_max)
{
    if (!out || !wv || !mu || !ga || !alpha || !beta || !gamma || !zeta) return SOSGPU_E_ARG;
    const int N = wv->n, B = wv->os_nb;
    if (N < 1 || N > 85 || B < 2 || B > 400 || iborm_max < 0 || iborm_max > B) return SOSGPU_E_ARG;
    if (wv->n0 < 1 || wv->n0 > N) return SOSGPU_E_ARG;          // the solar direction must be one of mu[]
    if (wv->igmax < 1) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));

    sosgpu_ctx *cx = new sosgpu_ctx();
    cx->device = device;
    cx->bytes = 0;
    cx->timed = false;
    cx->last_stream = nullptr;
    SosDev &d = cx->d;
    memset(&d, 0, sizeof(d));
    d.n = N; d.w = 2 * N + 1; d.r6 = 6 * N;
    d.kp = sos_round_up(6 * N, 8); d.ks2 = d.kp / 8;
    const int rt = (6 * N + 15) / 16;
    d.rtp = 4 * ((rt + 3) / 4);
    d.os_nb = B; d.smax = iborm_max;
    d.n0 = wv->n0; d.imat_surf = wv->imat_surf == 1; d.ifresnel = wv->ifresnel == 1 ? 1 : 0;
    d.igmax = wv->igmax; d.ipolar = wv->ipolar ? 1 : 0;
    d.mus = mu[wv->n0 - 1];
    d.ro = wv->ro;
    // molecular phase-matrix coefficients, SOS_OS.F:678-684
    double aaa = wv->ron / (2 - wv->ron);
    aaa = (1 - aaa) / (1 + 2 * aaa);
    d.beta2 = 0.5 * aaa; d.gamma2 = -aaa * sqrt(1.5); d.alpha2 = 3. * aaa;
    std::vector<double> coef((size_t)4 * (B + 1));
    for (int l = 0; l <= B; l++) {
        coef[l] = alpha[l]; coef[(B + 1) + l] = beta[l]; coef[2 * (B + 1) + l] = gamma[l]; coef[3 * (B + 1) + l] = zeta[l];
    }
    if (!d.ipolar) {    // SOS_OS.F:689-699
        d.gamma2 = 0.; d.alpha2 = 0.;
        for (int l = 0; l <= B; l++) { coef[l] = 0.; coef[2 * (B + 1) + l] = 0.; coef[3 * (B + 1) + l] = 0.; }
    }
    // thresholds: REAL*4 literals widened (SOS.h:389,394,400), D literal (SOS.h:395)
    d.thr_cv = (double)0.00001f; d.thr_sum = (double)0.00001f; d.thr_sf = (double)0.00001f; d.thr_val = 1.0e-50;
    // flat-sea Fresnel matrix, SOS_MAT_FRESNEL_PLAN_REFL (SOS_OS.F:1753-1780)
    std::vector<double> fres((size_t)3 * N, 0.);
    if (d.ifresnel) {
        for (int j = 0; j <= N; j++) {
            const double m = (j == 0) ? d.mus : mu[j - 1];
            const double ind2 = wv->ind_surf * wv->ind_surf, mu2 = m * m;
            const double x = sqrt(ind2 - 1.0 + mu2);
            const double rl = (ind2 * m - x) / (ind2 * m + x);
            const double rr = (m - x) / (m + x);
            const double f11 = (rl * rl + rr * rr) / 2.;
            const double f12 = d.ipolar ? (rl * rl - rr * rr) / 2. : 0.;
            const double f33 = d.ipolar ? rl * rr : 0.;
            if (j == 0) { d.f11sun = f11; d.f12sun = f12; }
            else { fres[j - 1] = f11; fres[N + j - 1] = f12; fres[2 * N + j - 1] = f33; }
        }
    }
    double *p;
    int rc;
#define UP(dst, src, cnt)                                                                     \
    if ((rc = dev_alloc(cx, &p, (cnt)))) { sosgpu_destroy(cx); return rc; }                   \
    if (hipMemcpy(p, (src), (cnt) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {   \
        sosgpu_destroy(cx); return SOSGPU_E_HIP; }                                            \
    dst = p;
    UP(d.mu, mu, (size_t)N)
    UP(d.ga, ga, (size_t)N)
    UP(d.coef, coef.data(), coef.size())
    UP(d.fres, fres.data(), fres.size())
#undef UP
    const size_t per = (size_t)d.rtp * d.ks2 * 128;
    if ((rc = dev_alloc(cx, &d.prt, (size_t)(d.smax + 1) * 3 * (B + 1) * d.w)) ||
        (rc = dev_alloc(cx, &d.mp_aer, (size_t)(d.smax + 1) * per)) ||
        (rc = dev_alloc(cx, &d.mp_ray, (size_t)3 * per)) ||
        (rc = dev_alloc(cx, &d.sv, (size_t)(d.smax + 1) * 4 * d.kp))) {
        sosgpu_destroy(cx);
        return rc;
    }
    hipMemset(d.mp_ray, 0, 3 * per * sizeof(double));
    if (hipEventCreate(&cx->ev0) != hipSuccess || hipEventCreate(&cx->ev1) != hipSuccess) {
        sosgpu_destroy(cx);
        return SOSGPU_E_HIP;
    }
    *out = cx;
    return SOSGPU_OK;
}

extern "C" int sosgpu_destroy(sosgpu_ctx *cx)
{
    if (!cx) return SOSGPU_OK;
    hipSetDevice(cx->device);
    for (void *p : cx->allocs) hipFree(p);
    if (cx->ev0) hipEventDestroy(cx->ev0);
    if (cx->ev1) hipEventDestroy(cx->ev1);
    delete cx;
    return SOSGPU_OK;
}

extern "C" size_t sosgpu_ctx_bytes(const sosgpu_ctx *cx) { return cx ? cx->bytes : 0; }

extern "C" int sosgpu_set_surface_matrices(sosgpu_ctx *cx, const float *d_rsurf)
{
    if (!cx) return SOSGPU_E_ARG;
    if (cx->d.imat_surf && !d_rsurf) return SOSGPU_E_ARG;
    cx->d.rsurf = d_rsurf;
    return SOSGPU_OK;
}

extern "C" int sosgpu_noyaux(sosgpu_ctx *cx, void *stream)
{
    if (!cx) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    launch_noyaux(cx->d, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}

extern "C" int sosgpu_noyaux_fetch(sosgpu_ctx *cx, int is, double *out)
{
    if (!cx || !out || is < 0 || is > cx->d.smax) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    const size_t cnt = (size_t)6 * cx->d.w * cx->d.w + 3 * cx->d.w;
    double *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, cnt * sizeof(double)));
    launch_noyaux_fetch(cx->d, is, tmp, nullptr);
    hipError_t e = hipMemcpy(out, tmp, cnt * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(e);
    return SOSGPU_OK;
}

extern "C" int sosgpu_os_solve(sosgpu_ctx *cx, int nb, int lp, const int32_t *d_nt, const int32_t *d_iborm,
                               const double *d_prof, const int32_t *d_jout, const double *d_zz,
                               double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux, void *stream)
{
    if (!cx || nb < 0 || lp < 2 || !d_nt || !d_iborm || !d_prof || !d_rec || !d_norders || !d_iglast || !d_flux)
        return SOSGPU_E_ARG;
    if ((d_jout == nullptr) != (d_zz == nullptr)) return SOSGPU_E_ARG;
    if (cx->d.imat_surf && !cx->d.rsurf) return SOSGPU_E_ARG;
    if (nb == 0) return SOSGPU_OK;
    HIPCHK(hipSetDevice(cx->device));
    SosBins bn;
    bn.nb = nb; bn.lp = lp; bn.nt = d_nt; bn.iborm = d_iborm; bn.jout = d_jout; bn.prof = d_prof; bn.zz = d_zz;
    bn.rec = d_rec; bn.flux = d_flux; bn.norders = d_norders; bn.iglast = d_iglast;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipEventRecord(cx->ev0, st));
    // lp - 1 bounds every NT of the batch (the host pads the level axis to lp)
    const int rc = launch_sos_os(cx->d, bn, lp - 1, st);
    if (rc == -2) { g_last_hip = (int)hipGetLastError(); return SOSGPU_E_HIP; }
    if (rc) return rc;
    HIPCHK(hipEventRecord(cx->ev1, st));
    cx->timed = true;
    cx->last_stream = st;
    return SOSGPU_OK;
}

extern "C" int sosgpu_last_solve_ms(sosgpu_ctx *cx, float *ms)
{
    if (!cx || !ms || !cx->timed) return SOSGPU_E_ARG;
    HIPCHK(hipEventSynchronize(cx->ev1));
    HIPCHK(hipEventElapsedTime(ms, cx->ev0, cx->ev1));
    return SOSGPU_OK;
}

extern "C" int sosgpu_os_flops(sosgpu_ctx *cx, int nb, const int32_t *d_nt, const int32_t *d_norders,
                               const int32_t *d_iglast, double *flops_out)
{
    if (!cx || !flops_out || nb < 0) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    const int S1 = cx->d.smax + 1;
    std::vector<int32_t> nt(nb), no(nb), ig((size_t)nb * S1);
    HIPCHK(hipMemcpy(nt.data(), d_nt, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(no.data(), d_norders, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ig.data(), d_iglast, (size_t)nb * S1 * sizeof(int32_t), hipMemcpyDeviceToHost));
    const double r6 = 6.0 * cx->d.n;
    double tot = 0.;
    for (int b = 0; b < nb; b++) {
        const double L = nt[b] + 1.0;
        for (int s = 0; s < no[b]; s++) {
            const int steps = ig[(size_t)b * S1 + s] - 1;    // scattering orders >= 2 actually computed
            if (steps <= 0) continue;
            double w = 2. * r6 * r6 * L + 12. * r6 * nt[b];  // SURVEY 8d W_step
            if (s <= 2) w += 2. * 3. * r6 * L * 3.;
            tot += steps * w;
        }
    }
    *flops_out = tot;
    return SOSGPU_OK;
}

extern "C" int sosgpu_aggregate(sosgpu_ctx *cx, int nseg, const int32_t *d_seg, const double *d_aik,
                                const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                                double *d_out_rec, double *d_out_scal, void *stream)
{
    if (!cx || nseg < 1 || !d_seg || !d_aik || !d_rec || !d_norders || !d_flux || !d_scal || !d_out_rec || !d_out_scal)
        return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    launch_aggregate(cx->d, nseg, d_seg, d_aik, d_rec, d_norders, d_flux, d_scal, d_out_rec, d_out_scal, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}
