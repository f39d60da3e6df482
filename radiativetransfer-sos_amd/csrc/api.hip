// csrc/api.hip -- extern "C" surface of libsosgpu.so (see include/sosgpu.h for the contract and the
// reference routine each entry point replaces).
#include "../../include/sosgpu.h"
#include "kernels.h"
#include "sos_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
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
    std::vector<size_t> alloc_cls;   // size class of each entry of allocs (mem_give)
    size_t bytes;
    hipEvent_t ev0, ev1;
    bool timed;
    hipStream_t last_stream;
    std::vector<hipStream_t> used_streams;   // every stream a solve / table build of this context was queued on
    std::vector<void *> host_staging;        // pinned host blocks of queued copies (sosgpu_ctx_table), freed by sosgpu_destroy
    int nt_max_hint;
    double ind_surf;
    unsigned long long *phase;   // diagnostic phase-cycle buffer (sosgpu_debug_phase_buffer), else null
    double *agg_partial;    // chunk partials of the large-batch aggregate
    double *scratch;        // field-in-HBM variant: grow-only per-bin scratch
    double *prof_ng;        // [4][608] no-gas profile of the wavelength (sosgpu_profile)
    hipStream_t prof_ng_stream;   // stream of the last sosgpu_profile (the block is rewritten in stream order)
    double *gnd_op, *gnd_dir;   // packed ground-reflection operators / solar-beam columns of the surface matrices (context-owned)
    size_t scratch_doubles;
    size_t dbg_spec_i3;     // offset of the order-parallel form's I3 block in the scratch of the last solve (diagnostic)
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
    case SOSGPU_E_RCCL: return "RCCL unavailable or an RCCL call failed";
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

// Device memory of the per-wavelength tables and of the entry points' temporaries comes from a process-wide pool of released
// blocks (size classes: 256 B steps up to 4 KiB, then eighths of the leading power of two; at most 16 GiB of the 288 / 8192 blocks kept:
// a chunk of sos_spectrum holds 256 contexts of six blocks each, and what the pool drops is a hipFree + hipMalloc per block).
// One sos_proc call = one context: without the pool every call pays ~10 hipMalloc + hipFree, and hipFree waits for the
// whole device -- i.e. for the kernels of every other host thread.  A block is only returned after the work that used it has been
// waited for (sosgpu_destroy: the context's streams; temporaries: their stream).  sosgpu_trim() empties the pool.
namespace {
struct PoolBlock { void *p; size_t n; int dev; };
std::mutex g_mem_mutex;
std::vector<PoolBlock> g_mem_free;
size_t g_mem_bytes = 0;

size_t mem_class(size_t n)
{
    if (n <= 4096) return (std::max<size_t>(n, 1) + 255) & ~(size_t)255;
    size_t p2 = 1;
    while ((p2 << 1) <= n) p2 <<= 1;
    const size_t step = p2 >> 3;
    return (n + step - 1) / step * step;
}

void mem_trim()
{
    std::vector<PoolBlock> all;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        all.swap(g_mem_free);
        g_mem_bytes = 0;
    }
    for (const PoolBlock &b : all) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); }
}

// *cls receives the size class (pass it back to mem_give); nullptr on failure
void *mem_take(int dev, size_t bytes, size_t *cls)
{
    const size_t c = mem_class(bytes);
    *cls = c;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        for (size_t i = g_mem_free.size(); i-- > 0;)
            if (g_mem_free[i].dev == dev && g_mem_free[i].n == c) {
                void *p = g_mem_free[i].p;
                g_mem_free.erase(g_mem_free.begin() + i);
                g_mem_bytes -= c;
                return p;
            }
    }
    void *p = nullptr;
    if (hipMalloc(&p, c) != hipSuccess) {
        (void)hipGetLastError();
        sosgpu_trim();                             // the pools themselves may be what fills the device: release and retry once
        (void)hipSetDevice(dev);
        if (hipMalloc(&p, c) != hipSuccess) return nullptr;
    }
    return p;
}

void mem_give(int dev, void *p, size_t cls)
{
    if (!p) return;
    std::vector<PoolBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        g_mem_free.push_back({p, cls, dev});
        g_mem_bytes += cls;
        while (g_mem_free.size() > 8192 || g_mem_bytes > ((size_t)16 << 30)) {     // oldest first
            g_mem_bytes -= g_mem_free.front().n;
            drop.push_back(g_mem_free.front());
            g_mem_free.erase(g_mem_free.begin());
        }
    }
    for (const PoolBlock &b : drop) (void)hipFree(b.p);
}

// temporary of one entry point: taken from the pool, returned by the destructor (the entry point has waited for its stream)
struct TmpBuf {
    int dev; void *p; size_t cls;
    TmpBuf(int device, size_t bytes) : dev(device), p(mem_take(device, bytes, &cls)) {}
    ~TmpBuf() { mem_give(dev, p, cls); }
    TmpBuf(const TmpBuf &) = delete;
    TmpBuf &operator=(const TmpBuf &) = delete;
};
}   // namespace

template <typename T>
static int dev_alloc(sosgpu_ctx *cx, T **p, size_t count)
{
    size_t cls = 0;
    void *q = mem_take(cx->device, count * sizeof(T), &cls);
    if (!q) { g_last_hip = (int)hipGetLastError(); return SOSGPU_E_HIP; }
    cx->allocs.push_back(q);
    cx->alloc_cls.push_back(cls);
    cx->bytes += count * sizeof(T);
    *p = static_cast<T *>(q);
    return 0;
}

// Ordering rule of the library (include/sosgpu.h, "Streams"): no entry point synchronises the DEVICE and none uses the
// null stream.  Host-synchronous entry points (sosgpu_create, sosgpu_ctx_table, sosgpu_os_flops, ...) move their data on a
// utility stream of the calling thread (non-blocking, created on first use, one per device) and wait for THAT stream only, so
// a call never waits for -- or is overtaken by -- work other host threads have queued on their own streams.
static hipStream_t util_stream(int device)
{
    static thread_local hipStream_t st[16] = {nullptr};
    if (device < 0 || device >= 16) return nullptr;
    if (!st[device]) {
        // highest priority: the runtime maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES), in order per queue, and a
        // host framework may have created dozens of streams (torch: 32 per priority) -- at normal priority this stream then shares
        // a queue with caller streams, and its 20 us copies wait behind their kernels (0.4 ms per sosgpu_create inside
        // sos_spectrum, whose side streams run 1-2 ms profile kernels).  High-priority streams have hardware queues of their own.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        if (hipStreamCreateWithPriority(&st[device], hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&st[device], hipStreamNonBlocking) != hipSuccess) st[device] = nullptr;
        }
    }
    return st[device];
}

// Wait for the (short) work just queued on a utility stream.  hipStreamSynchronize blocks the thread on an interrupt once its
// spin window is over -- with kernels of other streams in flight that costs 0.3-0.4 ms of wake-up latency for a 20 us copy
// (measured inside sos_spectrum: sosgpu_create 0.41 ms against 0.03 ms on an idle device) -- so poll the stream first.
static hipError_t wait_short(hipStream_t us)
{
    for (int i = 0; i < 4000; i++) {
        const hipError_t q = hipStreamQuery(us);
        if (q == hipSuccess) return hipSuccess;
        if (q != hipErrorNotReady) return q;
    }
    return hipStreamSynchronize(us);
}

static void note_stream(sosgpu_ctx *cx, hipStream_t st)
{
    cx->last_stream = st;
    if (std::find(cx->used_streams.begin(), cx->used_streams.end(), st) == cx->used_streams.end()) cx->used_streams.push_back(st);
}

// waits for everything this context has queued (its solves may be in flight on several streams of the caller)
static hipError_t sync_ctx_streams(sosgpu_ctx *cx)
{
    hipError_t e = hipSuccess;
    for (hipStream_t st : cx->used_streams) { const hipError_t r = hipStreamSynchronize(st); if (e == hipSuccess) e = r; }
    return e;
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
    cx->ind_surf = wv->ind_surf;
    cx->scratch = nullptr;
    cx->scratch_doubles = 0;
    cx->dbg_spec_i3 = 0;
    cx->phase = nullptr;
    cx->agg_partial = nullptr;
    cx->prof_ng = nullptr;
    cx->prof_ng_stream = nullptr;
    cx->gnd_op = nullptr;
    cx->gnd_dir = nullptr;
    SosDev &d = cx->d;
    memset(&d, 0, sizeof(d));
    d.n = N; d.w = 2 * N + 1; d.r6 = 6 * N;
    d.kp = sos_round_up(6 * N, 8);
    d.kh = sos_round_up(3 * N, 8);
    // half-system row order: directions with a non-zero quadrature weight first (component-major), the zero-weight
    // ones (the solar direction, user angles: SOS_ANGLES gives them weight 0) last.  Their operator COLUMNS vanish, so
    // the contraction only runs over K = 3 Nw (N = 41: 15 k-pairs instead of 16); their rows are still computed.
    int nwgt = 0;
    for (int j = 0; j < N; j++) nwgt += ga[j] != 0.0;
    if (nwgt < 1) { delete cx; return SOSGPU_E_ARG; }
    d.ks2h = (3 * nwgt + 7) / 8;
    {   // projection rows ride in the last quad of the last row tile when that quad is padding
        const int cand = ((3 * N - 1) / 16) * 16 + 12;
        d.prow = cand >= 3 * N ? cand : -1;
    }
    std::vector<int32_t> rowmap(d.kh, -1);
    {
        int pos = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int c = 0; c < 3; c++)
                for (int j = 0; j < N; j++)
                    if ((ga[j] != 0.0) == (pass == 0)) rowmap[pos++] = c * N + j;
    }
    d.rtph = (d.kh + 15) / 16;
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
    int rc;
    hipStream_t us = util_stream(device);
    if (!us) { delete cx; return SOSGPU_E_HIP; }
    // The context's small tables share ONE allocation, filled by ONE copy from a pinned block of the calling thread and one
    // memset: [mu | ga | coef | fres | rowmap] copied, [mp_vt | mp_uf] cleared (a spectrum creates one context per
    // wavelength: five pageable uploads, two fills and ten pool requests were 0.2 ms of each).  Offsets in doubles, each
    // a multiple of 32 (256-byte alignment of the operator fragments).
    auto up32 = [](size_t v) { return (v + 31) & ~(size_t)31; };
    const size_t o_mu = 0, o_ga = up32(o_mu + N), o_coef = up32(o_ga + N), o_fres = up32(o_coef + coef.size()),
                 o_map = up32(o_fres + fres.size()), o_vt = up32(o_map + (rowmap.size() + 1) / 2),
                 n_vt = (size_t)3 * d.ks2h * 128, n_uf = (size_t)3 * d.rtph * 64, o_uf = up32(o_vt + n_vt), n_small = o_uf + n_uf;
    // ... followed, in the same allocation, by the tables the kernels of sosgpu_noyaux / sosgpu_set_surface_matrices fill
    // (one pool request -- in a cold process one hipMalloc -- per context instead of six)
    const size_t per = (size_t)2 * d.rtph * d.ks2h * 128, S1c = (size_t)d.smax + 1;
    const size_t o_prt = up32(n_small), o_aer = up32(o_prt + S1c * 3 * (B + 1) * d.w), o_sv = up32(o_aer + S1c * per),
                 o_gop = up32(o_sv + S1c * 4 * d.kp), n_gop = d.imat_surf ? S1c * (per / 2) : 0, o_gdir = up32(o_gop + n_gop),
                 n_gdir = d.imat_surf ? S1c * 3 * N : 0, n_all = o_gdir + n_gdir;
    double *small = nullptr;
    if ((rc = dev_alloc(cx, &small, n_all))) { sosgpu_destroy(cx); return rc; }
    static thread_local double *stage = nullptr;           // (kept for the life of the thread: 64 KB at the largest N and OS_NB)
    static thread_local size_t stage_n = 0;
    if (stage_n < o_vt) {
        if (stage) (void)hipHostFree(stage);
        stage = nullptr; stage_n = 0;
        if (hipHostMalloc((void **)&stage, up32(o_vt + 4096) * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            g_last_hip = (int)hipGetLastError(); sosgpu_destroy(cx); return SOSGPU_E_HIP; }
        stage_n = up32(o_vt + 4096);
    }
    memset(stage, 0, o_vt * sizeof(double));
    memcpy(stage + o_mu, mu, (size_t)N * sizeof(double));
    memcpy(stage + o_ga, ga, (size_t)N * sizeof(double));
    memcpy(stage + o_coef, coef.data(), coef.size() * sizeof(double));
    memcpy(stage + o_fres, fres.data(), fres.size() * sizeof(double));
    memcpy(stage + o_map, rowmap.data(), rowmap.size() * sizeof(int32_t));
    d.mu = small + o_mu; d.ga = small + o_ga; d.coef = small + o_coef; d.fres = small + o_fres;
    d.rowmap = reinterpret_cast<int32_t *>(small + o_map);
    d.nwgt = nwgt;
    d.mp_vt = small + o_vt; d.mp_uf = small + o_uf;
    d.prt = small + o_prt; d.mp_aer = small + o_aer; d.sv = small + o_sv;
    if (d.imat_surf) { cx->gnd_op = small + o_gop; cx->gnd_dir = small + o_gdir; }
    // The upload and the fill run on the calling thread's utility stream and are waited for here (the pinned block is reused by
    // the thread's next call): when the call returns every table is in place, whatever stream sosgpu_noyaux and the solves are
    // queued on afterwards.  (Round 2 filled on the null stream: a non-blocking caller stream does not wait for it, and a fill
    // could land on top of the molecular operator sosgpu_noyaux had already packed.)
    if (hipMemcpyAsync(small, stage, o_vt * sizeof(double), hipMemcpyHostToDevice, us) != hipSuccess ||
        hipMemsetAsync(small + o_vt, 0, (n_small - o_vt) * sizeof(double), us) != hipSuccess ||
        wait_short(us) != hipSuccess) {
        g_last_hip = (int)hipGetLastError();
        sosgpu_destroy(cx);
        return SOSGPU_E_HIP;
    }
    cx->ev0 = cx->ev1 = nullptr;                           // (created by the first solve)
    *out = cx;
    return SOSGPU_OK;
}

// The streamed solver's scratch outlives its context: one sos_proc call = one context (one wavelength), and a fresh hipMalloc
// of the 60-1000 MB the order-parallel form wants costs ~9 ms per call (scripts/latency_bench.py: 1.1 ms solve, 9.2 ms first
// call) -- more than the solve.  Released buffers wait here (per device, at most 8 of them and 8 GiB in total) for the next
// context.  A buffer is only returned after the stream of its last solve has been synchronised.
namespace {
struct ScratchBuf { double *p; size_t n; int dev; };
std::mutex g_pool_mutex;
std::vector<ScratchBuf> g_pool;

double *pool_take(int dev, size_t need, size_t *got)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); i++)
            if (g_pool[i].dev == dev && g_pool[i].n >= need && (best < 0 || g_pool[i].n < g_pool[best].n)) best = i;
        if (best >= 0) {
            ScratchBuf b = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
            *got = b.n;
            return b.p;
        }
    }
    double *p = nullptr;
    if (getenv("SOSGPU_DEBUG_POOL")) fprintf(stderr, "[sosgpu pool] hipMalloc %.1f MB\n", need * 8e-6);
    if (hipMalloc((void **)&p, need * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        sosgpu_trim();                             // up to 8 GiB of kept scratch + 16 GiB of kept tables: release, retry once
        (void)hipSetDevice(dev);
        if (hipMalloc((void **)&p, need * sizeof(double)) != hipSuccess) return nullptr;
    }
    *got = need;
    return p;
}

void pool_give(int dev, double *p, size_t n)
{
    if (!p) return;
    std::vector<double *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        g_pool.push_back({p, n, dev});
        size_t total = 0;
        for (const ScratchBuf &b : g_pool) total += b.n * sizeof(double);
        while (g_pool.size() > 8 || total > ((size_t)8 << 30)) {     // oldest first
            total -= g_pool.front().n * sizeof(double);
            drop.push_back(g_pool.front().p);
            g_pool.erase(g_pool.begin());
        }
    }
    if (!drop.empty() && getenv("SOSGPU_DEBUG_POOL")) fprintf(stderr, "[sosgpu pool] hipFree of %zu buffer(s)\n", drop.size());
    for (double *q : drop) (void)hipFree(q);
}
}   // namespace

extern "C" int sosgpu_trim(void)
{
    std::vector<ScratchBuf> all;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        all.swap(g_pool);
    }
    for (const ScratchBuf &b : all) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); }
    mem_trim();
    return SOSGPU_OK;
}

extern "C" int sosgpu_destroy(sosgpu_ctx *cx)
{
    if (!cx) return SOSGPU_OK;
    // teardown: nothing useful can be done with a failing free, errors are deliberately dropped
    (void)hipSetDevice(cx->device);
    (void)sync_ctx_streams(cx);        // solves / table builds of this context may still be running, on any of its streams
    for (void *h : cx->host_staging) (void)hipHostFree(h);
    for (size_t i = 0; i < cx->allocs.size(); i++) mem_give(cx->device, cx->allocs[i], cx->alloc_cls[i]);
    if (cx->scratch) pool_give(cx->device, cx->scratch, cx->scratch_doubles);
    if (cx->ev0) (void)hipEventDestroy(cx->ev0);
    if (cx->ev1) (void)hipEventDestroy(cx->ev1);
    delete cx;
    return SOSGPU_OK;
}

extern "C" size_t sosgpu_ctx_bytes(const sosgpu_ctx *cx) { return cx ? cx->bytes : 0; }

// Ground-reflection operator of a BRDF/BPDF surface in the packed A-fragment layout of the source operators (one system,
// rows = up-going half-system positions (c, k), columns = down-going positions (b, j), sos_common.h):
//   G[(c,k)][(b,j)] = (2/mu_k) w_j R_cb(j, k)   (SOS_OS.F:1194-1220; R_cb(I = j, J = k) = r[s][c*3+b][k*N + j])
//                     + 2 rho w_j mu_j for c = b = 0 and s = 0 (Lambertian part, SOS_OS.F:1177-1190)
// with the polarisation cut of SOS_OS.F:928-941 (IPOLAR = 0: only R_11 is kept), plus the solar-beam column
// rdir[s][c][k] = R_c1(N0, k) of the direct term (SOS_OS.F:984-990).
__global__ void k_pack_ground(SosDev cx, const float *__restrict__ r, double *__restrict__ gop, double *__restrict__ rdir)
{
    const int s = blockIdx.y, N = cx.n;
    const size_t per = (size_t)cx.rtph * cx.ks2h * 128;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float *rs = r + (size_t)s * 9 * N * N;
    if (q < per) {
        const int e2 = q & 1, lane = (q >> 1) & 63;
        const int m = (int)((q >> 7) % cx.ks2h), rt = (int)((q >> 7) / cx.ks2h);
        // A-operand order of v_mfma_f64_4x4x4f64 (sos_dev.h ground_mfma): block = row quad, lane>>4 = k
        const int row = rt * 16 + ((lane >> 2) & 3) * 4 + (lane & 3), col = 8 * m + 4 * e2 + (lane >> 4);
        double v = 0.;
        if (row < 3 * N && col < 3 * N) {
            const int ro = cx.rowmap[row], co = cx.rowmap[col];
            const int c = ro / N, k = ro % N, b = co / N, j = co % N;
            double x = rs[(size_t)(c * 3 + b) * N * N + (size_t)k * N + j];
            if (!cx.ipolar && (c || b)) x = 0.;
            v = (2. / cx.mu[k]) * cx.ga[j] * x;
            if (c == 0 && b == 0 && s == 0 && cx.ro != 0.) v = v + 2. * cx.ro * cx.ga[j] * cx.mu[j];
        }
        gop[(size_t)s * per + q] = v;
    }
    if (q < (size_t)3 * N) {
        const int c = (int)(q / N), k = (int)(q % N);
        double x = rs[(size_t)(c * 3) * N * N + (size_t)k * N + (cx.n0 - 1)];
        if (!cx.ipolar && c) x = 0.;
        rdir[(size_t)s * 3 * N + q] = x;
    }
}

static int set_surface_matrices_impl(sosgpu_ctx *cx, const float *d_rsurf, hipStream_t st)
{
    if (!cx) return SOSGPU_E_ARG;
    if (cx->d.imat_surf && !d_rsurf) return SOSGPU_E_ARG;
    if (!d_rsurf) { cx->d.mp_gnd = nullptr; cx->d.rdir = nullptr; return SOSGPU_OK; }
    HIPCHK(hipSetDevice(cx->device));
    const size_t per = (size_t)cx->d.rtph * cx->d.ks2h * 128, S1 = (size_t)cx->d.smax + 1;
    if (!cx->gnd_op) {
        int rc = dev_alloc(cx, &cx->gnd_op, S1 * per);
        if (!rc) rc = dev_alloc(cx, &cx->gnd_dir, S1 * 3 * cx->d.n);
        if (rc) return rc;
    }
    dim3 grid((unsigned)((std::max(per, (size_t)3 * cx->d.n) + 255) / 256), (unsigned)S1);
    k_pack_ground<<<grid, 256, 0, st>>>(cx->d, d_rsurf, cx->gnd_op, cx->gnd_dir);
    HIPCHK(hipGetLastError());
    note_stream(cx, st);
    cx->d.mp_gnd = cx->gnd_op;
    cx->d.rdir = cx->gnd_dir;
    return SOSGPU_OK;
}

// stream-ordered form: the packing kernel is queued on `stream` (where d_rsurf was produced, or after it was complete) and
// nothing is waited for; d_rsurf must stay allocated until that work has run
extern "C" int sosgpu_set_surface_matrices_async(sosgpu_ctx *cx, const float *d_rsurf, void *stream)
{
    return set_surface_matrices_impl(cx, d_rsurf, (hipStream_t)stream);
}

// host-synchronous form (round 1's contract: the caller may release d_rsurf on return).  d_rsurf must be complete when the call
// is made -- the packing runs on the calling thread's utility stream, which is all the call waits for.
extern "C" int sosgpu_set_surface_matrices(sosgpu_ctx *cx, const float *d_rsurf)
{
    if (!cx) return SOSGPU_E_ARG;
    hipStream_t us = util_stream(cx->device);
    if (!us) return SOSGPU_E_HIP;
    const int rc = set_surface_matrices_impl(cx, d_rsurf, us);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(us));
    return SOSGPU_OK;
}

extern "C" int sosgpu_noyaux(sosgpu_ctx *cx, void *stream)
{
    if (!cx) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    launch_noyaux(cx->d, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    note_stream(cx, (hipStream_t)stream);
    return SOSGPU_OK;
}

extern "C" int sosgpu_noyaux_fetch(sosgpu_ctx *cx, int is, double *out)
{
    if (!cx || !out || is < 0 || is > cx->d.smax) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    const size_t cnt = (size_t)6 * cx->d.w * cx->d.w + 3 * cx->d.w;
    hipStream_t us = util_stream(cx->device);
    if (!us) return SOSGPU_E_HIP;
    TmpBuf tmp(cx->device, cnt * sizeof(double));
    if (!tmp.p) return SOSGPU_E_HIP;
    HIPCHK(sync_ctx_streams(cx));            // sosgpu_noyaux may still be running on the stream it was queued on
    launch_noyaux_fetch(cx->d, is, (double *)tmp.p, us);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, tmp.p, cnt * sizeof(double), hipMemcpyDeviceToHost, us));
    HIPCHK(hipStreamSynchronize(us));
    return SOSGPU_OK;
}

// sosgpu_os_solve (table == null) and sosgpu_os_solve_multi (per-bin contexts from a device table)
static int os_solve_impl(sosgpu_ctx *cx, const SosDev *table, const int32_t *d_ctx_of_bin, const int32_t *d_order, int nb, int lp, const int32_t *d_nt,
                         const int32_t *d_iborm, const double *d_prof, const int32_t *d_jout, const double *d_zz,
                         double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux, void *stream)
{
    if (!cx || nb < 0 || lp < 2 || !d_nt || !d_iborm || !d_prof || !d_rec || !d_norders || !d_iglast || !d_flux)
        return SOSGPU_E_ARG;
    if ((d_jout == nullptr) != (d_zz == nullptr)) return SOSGPU_E_ARG;
    if (cx->d.imat_surf && !cx->d.mp_gnd) return SOSGPU_E_ARG;
    if (nb == 0) return SOSGPU_OK;
    HIPCHK(hipSetDevice(cx->device));
    hipStream_t st = (hipStream_t)stream;
    // variant: field in LDS, or (NT too large) field in a per-bin HBM scratch, launched in sub-batches so that
    // the scratch stays below its budget
    int nw, rtw, ct, big;
    size_t lds;
    const int nt_max = lp - 1;          // lp - 1 bounds every NT of the batch (the host pads the level axis to lp)
    int rc = sos_os_variant(cx->d.n, nt_max, &nw, &rtw, &ct, &lds, &big);
    if (rc) return rc;
    int per_launch = nb, spec_k = 0;
    int lpb = sos_round_up(lp, 32);
    size_t per_bin = big ? sos_stream_scratch_doubles(cx->d.n, lpb) : 0;
    if (big) {
        // scratch budget: 64 GiB of the 288 GB (a launch covers ~40 000 bins at 608 levels; SOSGPU_SCRATCH_GIB overrides it)
        size_t gib = 64;
        if (const char *e = getenv("SOSGPU_SCRATCH_GIB")) { const long v = atol(e); if (v > 0) gib = (size_t)v; }
        const size_t cap = (gib << 30) / sizeof(double);
        per_launch = (int)std::min<size_t>((size_t)nb, std::max<size_t>(1, cap / per_bin));
        // Few bins (a band of one wavelength): the order-parallel form -- up to 48 Fourier orders of every bin at a time, each in a
        // work region of its own, so that the band fills ~1024 workgroup slots (sos_stream.hip; SOSGPU_STREAM_SPEC=0 turns it
        // off, SOSGPU_STREAM_SPEC_MAXBINS moves the limit).  A single bin takes 11 ms as one workgroup, ~1.5 ms this way.
        int spec_max = 128;
        if (const char *e = getenv("SOSGPU_STREAM_SPEC")) { if (atoi(e) == 0) spec_max = 0; }
        if (const char *e = getenv("SOSGPU_STREAM_SPEC_MAXBINS")) spec_max = atoi(e);
        if (const char *e = getenv("SOSGPU_STREAM_PERSIST")) { if (atoi(e) != 0) spec_max = 0; }      // explicitly chosen forms win
        if (const char *e = getenv("SOSGPU_STREAM_ORDERS_PER_LAUNCH")) { if (atoi(e) > 0) spec_max = 0; }
        const int s1n = cx->d.smax + 1;
        if (!table && nb <= spec_max && s1n > 1) {
            // first round: 48 orders (a series typically ends after 25-50 of its up to 81), 24 above 40 bins; later rounds run
            // half as many.  Measured (profiles/r02_sos_proc_latency.txt): the number of rounds is what costs, not the tasks
            // beyond the chip's 512 workgroup slots.
            spec_k = std::min(s1n, nb <= 40 ? 48 : 24);
            // The work regions are laid out for the batch's own level count, not for the padded row length `lp` of the caller
            // (608 for profiles made by sosgpu_profile): the few NT come to the host -- one small copy behind the work already
            // queued on the stream, which this latency-bound form has to wait for anyway.
            std::vector<int32_t> h_nt((size_t)nb);
            HIPCHK(hipMemcpyAsync(h_nt.data(), d_nt, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            int ntm = 1;
            for (int b = 0; b < nb; b++) if (h_nt[b] < lp) ntm = std::max(ntm, (int)h_nt[b]);    // (malformed bins are flagged by the kernel)
            lpb = std::min(lpb, sos_round_up(ntm + 1, 32));
            per_bin = sos_stream_scratch_doubles(cx->d.n, lpb);
            // at most 4 GiB of work regions (128 bins x 24 orders at 600 levels need 4.6): beyond, fewer orders per round
            const size_t soft = ((size_t)4 << 30) / sizeof(double);
            const size_t fit = soft / ((size_t)nb * per_bin);
            if (fit < (size_t)spec_k) spec_k = std::max(std::min(spec_k, 8), (int)fit);
            if (const char *e = getenv("SOSGPU_STREAM_SPEC_K")) spec_k = std::min(s1n, std::max(1, atoi(e)));   // (tests)
            if ((size_t)nb * spec_k * per_bin > cap) spec_k = 0;
            if (!spec_k) { lpb = sos_round_up(lp, 32); per_bin = sos_stream_scratch_doubles(cx->d.n, lpb); }
        }
        const size_t regions = spec_k ? (size_t)nb * spec_k : (size_t)per_launch;
        const size_t i3_doubles = spec_k ? (size_t)nb * s1n * sos_stream_threads(cx->d.n) : 0;
        // + the task queues and per-bin order flags of the persistent form (ints, behind the bins' scratch)
        const size_t need = per_bin * regions + i3_doubles + (256 + (size_t)per_launch) / 2 + 1;
        if (need > cx->scratch_doubles) {
            if (cx->scratch) {
                HIPCHK(sync_ctx_streams(cx));   // an earlier solve of this context may still be using it, on any of its streams
                pool_give(cx->device, cx->scratch, cx->scratch_doubles);
            }
            cx->scratch = nullptr;
            cx->scratch_doubles = 0;
            size_t got = 0;
            cx->scratch = pool_take(cx->device, need, &got);
            if (!cx->scratch) { g_last_hip = (int)hipGetLastError(); return SOSGPU_E_HIP; }
            cx->scratch_doubles = got;
        }
    }
    // (The scratch is reused from solve to solve and from context to context -- pool above -- without being cleared: the streamed
    //  kernel initialises what it reads; the pad levels and pad columns it stages along with a chunk feed columns / rows of
    //  the contraction that are never stored.)
    if (!cx->ev0) { HIPCHK(hipEventCreate(&cx->ev0)); }
    if (!cx->ev1) { HIPCHK(hipEventCreate(&cx->ev1)); }
    HIPCHK(hipEventRecord(cx->ev0, st));
    const int S1 = cx->d.smax + 1, W = cx->d.w;
    for (int b0 = 0; b0 < nb; b0 += per_launch) {
        SosBins bn;
        bn.nb = std::min(per_launch, nb - b0); bn.lp = lp;
        bn.nt = d_nt + b0; bn.iborm = d_iborm + b0; bn.jout = d_jout ? d_jout + b0 : nullptr;
        bn.prof = d_prof + (size_t)b0 * 3 * lp; bn.zz = d_zz ? d_zz + b0 : nullptr;
        bn.rec = d_rec + (size_t)b0 * S1 * 3 * W; bn.flux = d_flux + (size_t)2 * b0;
        bn.norders = d_norders + b0; bn.iglast = d_iglast + (size_t)b0 * S1;
        bn.scratch = big ? cx->scratch : nullptr; bn.scr_stride = per_bin; bn.lpb = lpb;
        bn.phase = cx->phase ? cx->phase + (size_t)b0 * 8 : nullptr;
        bn.ctxs = table; bn.ctx_of_bin = table ? d_ctx_of_bin + b0 : nullptr;
        bn.order = (table && per_launch >= nb) ? d_order : nullptr;      // (a split batch keeps its given order)
        bn.queue = bn.qflag = nullptr; bn.q_tail = -1;
        bn.spec_k = 0; bn.spec_i3 = nullptr;
        if (const char *e = getenv("SOSGPU_STREAM_QTAIL")) bn.q_tail = atoi(e);
        bn.s_begin = 0; bn.s_end = S1;
        if (big && spec_k) {
            // order-parallel form: set-up launch, then rounds of (K order tasks per bin, replay of their stop tests)
            bn.spec_i3 = cx->scratch + per_bin * (size_t)nb * spec_k;
            cx->dbg_spec_i3 = per_bin * (size_t)nb * spec_k;
            const int nt_max_r = lpb - 1;              // level capacity of the regions (>= every valid NT of the batch)
            bn.spec_k = -spec_k; bn.s_begin = 0; bn.s_end = 0;
            rc = launch_sos_stream(cx->d, bn, nt_max_r, st, &g_last_hip);
            bn.spec_k = spec_k;
            // (a series typically ends after 25-50 of its up to 81 orders: 32 + 16 + ... wastes less than all at once, and a
            //  launch whose bins have all stopped costs a few microseconds)
            for (int s0 = 0, kr = spec_k; s0 < S1 && rc == 0; s0 += kr, kr = std::max(1, spec_k / 2)) {
                bn.s_begin = s0; bn.s_end = std::min(S1, s0 + kr);
                rc = launch_sos_stream(cx->d, bn, nt_max_r, st, &g_last_hip);
                if (rc == 0) rc = launch_sos_stream_replay(cx->d, bn, bn.s_begin, bn.s_end, st, &g_last_hip);
            }
        } else if (big) {
            // The streamed kernel can run `opl` Fourier orders of every bin per launch (order-synchronous launches: every
            // workgroup then streams the same source operator).  Measured on the realistic mix (profiles/r02_stream_experiments.txt):
            // 1 order per launch 21.1k bins/s, all orders in one launch 21.9k -- the operator stream is not what binds, so one
            // launch is the default for large batches; SOSGPU_STREAM_ORDERS_PER_LAUNCH = n selects n orders per launch (tests cover both).
            int opl = 0;
            if (const char *e = getenv("SOSGPU_STREAM_ORDERS_PER_LAUNCH")) opl = atoi(e);
            if (opl <= 0) opl = S1;
            // SOSGPU_STREAM_PERSIST=1: ONE persistent launch whose workgroups take (Fourier order, bin) tasks from per-XCD queues,
            // so that the workgroups of an XCD share the source operators of one or two orders in L2 (sos_stream.hip, PERSIST).
            // Measured on the realistic mix (profiles/r02_stream_experiments.txt): fabric reads 505 -> 317 GB per launch (the
            // operator misses are gone), 22.2 k against 22.6 k bins/s -- the kernel is not bound by that traffic, so one
            // workgroup per bin stays the default.  Launches with a context table always use the default form.
            int persist = 0;
            if (const char *e = getenv("SOSGPU_STREAM_PERSIST")) persist = atoi(e);
            if (persist && !table && opl == S1) {
                bn.queue = reinterpret_cast<int *>(cx->scratch + per_bin * per_launch);
                bn.qflag = bn.queue + 256;
                HIPCHK(hipMemsetAsync(bn.queue, 0, (256 + (size_t)bn.nb) * sizeof(int), st));
            }
            for (int s0 = 0; s0 < S1 && rc == 0; s0 += opl) {
                bn.s_begin = s0; bn.s_end = std::min(S1, s0 + opl);
                rc = table ? launch_sos_stream_multi(cx->d, bn, nt_max, st, &g_last_hip)
                           : launch_sos_stream(cx->d, bn, nt_max, st, &g_last_hip);
            }
        } else rc = table ? launch_sos_os_multi(cx->d, bn, nt_max, st, &g_last_hip)
                          : launch_sos_os(cx->d, bn, nt_max, st, &g_last_hip);
        if (rc == -2) return SOSGPU_E_HIP;
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(cx->ev1, st));
    cx->timed = true;
    note_stream(cx, st);
    return SOSGPU_OK;
}

extern "C" int sosgpu_os_solve(sosgpu_ctx *cx, int nb, int lp, const int32_t *d_nt, const int32_t *d_iborm,
                               const double *d_prof, const int32_t *d_jout, const double *d_zz,
                               double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux, void *stream)
{
    return os_solve_impl(cx, nullptr, nullptr, nullptr, nb, lp, d_nt, d_iborm, d_prof, d_jout, d_zz, d_rec, d_norders, d_iglast,
                         d_flux, stream);
}

extern "C" size_t sosgpu_ctx_table_entry_bytes(void) { return sizeof(SosDev); }

extern "C" int sosgpu_ctx_table(sosgpu_ctx *const *ctxs, int nctx, void *d_table, void *stream)
{
    if (!ctxs || nctx < 1 || !d_table || !ctxs[0]) return SOSGPU_E_ARG;
    const SosDev &a = ctxs[0]->d;
    // The table is copied on the caller's stream -- d_table is the caller's memory and may still be read by work queued there
    // (a buffer the caller's allocator has just recycled) -- from a pinned staging block the first context keeps until it is
    // destroyed, so nothing is waited for.
    HIPCHK(hipSetDevice(ctxs[0]->device));
    SosDev *tab = nullptr;
    HIPCHK(hipHostMalloc((void **)&tab, (size_t)nctx * sizeof(SosDev), hipHostMallocDefault));
    ctxs[0]->host_staging.push_back(tab);
    for (int i = 0; i < nctx; i++) {
        if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device) return SOSGPU_E_ARG;
        const SosDev &d = ctxs[i]->d;
        // what selects the kernel variant and the record layout must agree over the table
        if (d.n != a.n || d.kh != a.kh || d.ks2h != a.ks2h || d.rtph != a.rtph || d.smax != a.smax ||
            (d.imat_surf != 0) != (a.imat_surf != 0))
            return SOSGPU_E_ARG;
        if (d.imat_surf && !d.mp_gnd) return SOSGPU_E_ARG;
        tab[i] = d;
    }
    HIPCHK(hipMemcpyAsync(d_table, tab, (size_t)nctx * sizeof(SosDev), hipMemcpyHostToDevice, (hipStream_t)stream));
    note_stream(ctxs[0], (hipStream_t)stream);
    return SOSGPU_OK;
}

extern "C" int sosgpu_os_solve_multi(sosgpu_ctx *cx, const void *d_table, const int32_t *d_ctx_of_bin, const int32_t *d_order,
                                     int nb, int lp,
                                     const int32_t *d_nt, const int32_t *d_iborm, const double *d_prof, const int32_t *d_jout,
                                     const double *d_zz, double *d_rec, int32_t *d_norders, int32_t *d_iglast, double *d_flux,
                                     void *stream)
{
    if (!d_table || !d_ctx_of_bin) return SOSGPU_E_ARG;
    return os_solve_impl(cx, static_cast<const SosDev *>(d_table), d_ctx_of_bin, d_order, nb, lp, d_nt, d_iborm, d_prof, d_jout,
                         d_zz, d_rec, d_norders, d_iglast, d_flux, stream);
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
    hipStream_t us = util_stream(cx->device);
    if (!us) return SOSGPU_E_HIP;
    HIPCHK(sync_ctx_streams(cx));              // the solve whose counts are read
    HIPCHK(hipMemcpyAsync(nt.data(), d_nt, nb * sizeof(int32_t), hipMemcpyDeviceToHost, us));
    HIPCHK(hipMemcpyAsync(no.data(), d_norders, nb * sizeof(int32_t), hipMemcpyDeviceToHost, us));
    HIPCHK(hipMemcpyAsync(ig.data(), d_iglast, (size_t)nb * S1 * sizeof(int32_t), hipMemcpyDeviceToHost, us));
    HIPCHK(hipStreamSynchronize(us));
    const double r6 = 6.0 * cx->d.n, r3 = 3.0 * cx->d.n, k3 = 3.0 * cx->d.nwgt;
    double tot = 0., exe = 0.;
    for (int b = 0; b < nb; b++) {
        const double L = nt[b] + 1.0;
        for (int s = 0; s < no[b]; s++) {
            const int steps = ig[(size_t)b * S1 + s] - 1;    // scattering orders >= 2 actually computed
            if (steps <= 0) continue;
            double w = 2. * r6 * r6 * L + 12. * r6 * nt[b];  // SURVEY 8d W_step
            double e = 2. * 2. * r3 * k3 * L + 10. * r6 * nt[b];
            if (s <= 2) { w += 2. * 3. * r6 * L * 3.; e += 2. * 4. * (r3 + k3) * L; }
            tot += steps * w;
            exe += steps * e;
        }
    }
    flops_out[0] = tot;
    flops_out[1] = exe;
    return SOSGPU_OK;
}

__global__ void k_aggregate_empty(int nel, int sw, double *out_rec, double *out_scal)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nel) out_rec[e] = 0.;
    if (e < sw) out_scal[e] = (e == 8) ? -2147483647. : 0.;
}

extern "C" int sosgpu_aggregate(sosgpu_ctx *cx, int nb, int nseg, const int32_t *d_seg, const double *d_aik,
                                const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                                const double *d_tdifmug, double *d_out_rec, double *d_out_scal, void *stream)
{
    if (!cx || nb < 0 || nseg < 1 || !d_out_rec || !d_out_scal) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    hipStream_t st = (hipStream_t)stream;
    const int nel = (cx->d.smax + 1) * 3 * cx->d.w;
    if (nb == 0) {          // empty shard of a band: neutral element of the cross-rank reduce
        if (nseg != 1) return SOSGPU_E_ARG;
        const int sw = SOSGPU_SCAL_BASE + cx->d.n;
        k_aggregate_empty<<<(std::max(nel, sw) + 255) / 256, 256, 0, st>>>(nel, sw, d_out_rec, d_out_scal);
        HIPCHK(hipGetLastError());
        return SOSGPU_OK;
    }
    if (nseg > nb || !d_seg || !d_aik || !d_rec || !d_norders || !d_flux || !d_scal) return SOSGPU_E_ARG;
    const int max_chunks = 4096;
    const int nb_single = (nseg == 1) ? nb : 0;      // one band: big batches use the chunked reduction
    if (nb_single > 128 && !cx->agg_partial) {
        if (int rc = dev_alloc(cx, &cx->agg_partial, (size_t)nel * max_chunks)) return rc;
    }
    launch_aggregate(cx->d, nseg, d_seg, d_aik, d_rec, d_norders, d_flux, d_scal, d_tdifmug, d_out_rec, d_out_scal, st,
                     nb_single, cx->agg_partial, max_chunks);
    HIPCHK(hipGetLastError());
    note_stream(cx, st);
    return SOSGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Cross-GPU reduce of the band partials over RCCL (xGMI).  librccl is resolved lazily with dlopen: a process that
// already holds an RCCL (torch's) gets that one, single-GPU users never load it.
// ---------------------------------------------------------------------------------------------
struct Uid { char internal[SOSGPU_UNIQUE_ID_BYTES]; };   // ncclUniqueId (nccl.h: 128 opaque bytes, passed by value)
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Uid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
}  // namespace
static Rccl g_rccl;
static int rccl_load_once()
{
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return SOSGPU_E_RCCL;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart ||
        !g_rccl.GroupEnd)
        return SOSGPU_E_RCCL;
    g_rccl.h = h;
    return 0;
}
static int rccl_load()
{
    static std::once_flag once;            // host threads may reach the first RCCL call together
    static int rc = SOSGPU_E_RCCL;
    std::call_once(once, [] { rc = rccl_load_once(); });
    return rc;
}

extern "C" int sosgpu_comm_unique_id(char id[SOSGPU_UNIQUE_ID_BYTES])
{
    if (!id) return SOSGPU_E_ARG;
    if (int rc = rccl_load()) return rc;
    Uid u;
    memset(&u, 0, sizeof u);
    if (g_rccl.GetUniqueId(&u) != 0) return SOSGPU_E_RCCL;
    memcpy(id, u.internal, SOSGPU_UNIQUE_ID_BYTES);
    return SOSGPU_OK;
}

extern "C" int sosgpu_comm_init_rank(void **comm, int nranks, const char id[SOSGPU_UNIQUE_ID_BYTES], int rank)
{
    if (!comm || !id || nranks < 1 || rank < 0 || rank >= nranks) return SOSGPU_E_ARG;
    if (int rc = rccl_load()) return rc;
    Uid u;
    memcpy(u.internal, id, SOSGPU_UNIQUE_ID_BYTES);
    return g_rccl.CommInitRank(comm, nranks, u, rank) == 0 ? SOSGPU_OK : SOSGPU_E_RCCL;
}

extern "C" int sosgpu_comm_destroy(void *comm)
{
    if (!comm) return SOSGPU_OK;
    if (int rc = rccl_load()) return rc;
    return g_rccl.CommDestroy(comm) == 0 ? SOSGPU_OK : SOSGPU_E_RCCL;
}

extern "C" int sosgpu_pack(sosgpu_ctx *cx, int nseg, const double *d_out_rec, const double *d_out_scal, double *d_buf, void *stream)
{
    if (!cx || nseg < 1 || !d_out_rec || !d_out_scal || !d_buf) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    const size_t nel = (size_t)(cx->d.smax + 1) * 3 * cx->d.w, sw = SOSGPU_SCAL_BASE + cx->d.n, row = nel + sw;
    HIPCHK(hipMemcpy2DAsync(d_buf, row * 8, d_out_rec, nel * 8, nel * 8, nseg, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIPCHK(hipMemcpy2DAsync(d_buf + nel, row * 8, d_out_scal, sw * 8, sw * 8, nseg, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SOSGPU_OK;
}

extern "C" int sosgpu_unpack(sosgpu_ctx *cx, int nseg, const double *d_buf, double *d_out_rec, double *d_out_scal, void *stream)
{
    if (!cx || nseg < 1 || !d_out_rec || !d_out_scal || !d_buf) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    const size_t nel = (size_t)(cx->d.smax + 1) * 3 * cx->d.w, sw = SOSGPU_SCAL_BASE + cx->d.n, row = nel + sw;
    HIPCHK(hipMemcpy2DAsync(d_out_rec, nel * 8, d_buf, row * 8, nel * 8, nseg, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIPCHK(hipMemcpy2DAsync(d_out_scal, sw * 8, d_buf + nel, row * 8, sw * 8, nseg, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SOSGPU_OK;
}

// the two MAX-combined scalars of every segment are saved before the SUM all-reduce and restored from their own MAX
// all-reduce afterwards
__global__ void k_reduce_save(int nseg, size_t row, size_t off, double *buf, double *mx, int restore)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * nseg) return;
    double *p = buf + (size_t)(i / 2) * row + off + 7 + (i & 1);
    if (restore) *p = mx[i]; else mx[i] = *p;
}

extern "C" int sosgpu_reduce(sosgpu_ctx *cx, void *comm, int nseg, double *d_buf, void *stream)
{
    if (!cx || nseg < 1 || !d_buf) return SOSGPU_E_ARG;
    if (!comm) return SOSGPU_OK;                  // single rank
    if (int rc = rccl_load()) return rc;
    HIPCHK(hipSetDevice(cx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nel = (size_t)(cx->d.smax + 1) * 3 * cx->d.w, row = nel + SOSGPU_SCAL_BASE + cx->d.n;
    TmpBuf mxb(cx->device, (size_t)2 * nseg * sizeof(double));
    if (!mxb.p) return SOSGPU_E_HIP;
    double *mx = (double *)mxb.p;
    k_reduce_save<<<(2 * nseg + 63) / 64, 64, 0, st>>>(nseg, row, nel, d_buf, mx, 0);
    const int ncclDouble = 8, ncclSum = 0, ncclMax = 2;    // nccl.h enumerators (ncclFloat64 = 8)
    int bad = g_rccl.AllReduce(d_buf, d_buf, row * nseg, ncclDouble, ncclSum, comm, st);
    bad |= g_rccl.AllReduce(mx, mx, (size_t)2 * nseg, ncclDouble, ncclMax, comm, st);
    k_reduce_save<<<(2 * nseg + 63) / 64, 64, 0, st>>>(nseg, row, nel, d_buf, mx, 1);
    hipError_t e = hipStreamSynchronize(st);
    if (bad) return SOSGPU_E_RCCL;
    HIPCHK(e);
    return SOSGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Cox-Munk glitter: host part = SOS_MAT_FRESNEL (SOS_SURFACE.F:1235-1603), O(N*OS_NS) flops and a
// 4(E15.8) text round trip (:1552 write, :1822 read) that has to be reproduced digit for digit.
// ---------------------------------------------------------------------------------------------
static double through_e15_8(double x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.7E", x);   // 8 significant digits, round to nearest = Fortran E15.8
    return strtod(buf, nullptr);
}

static double sigma2_of_wind(double wind) { return (double)0.003f + (double)0.00512f * wind; }   // SOS_GLITTER.F:300

extern "C" int sosgpu_mat_fresnel_host(int n, const double *mu, const double *chr, double ind, int os_ns, double *out)
{
    if (n < 1 || os_ns < 2 || !mu || !chr || !out) return SOSGPU_E_ARG;
    const int K = os_ns + 1;
    double *alpha = out, *beta = out + K, *gamma = out + 2 * K, *zeta = out + 3 * K;
    std::vector<double> delta(K, 0.), pl(os_ns + 3, 0.), pol(os_ns + 2, 0.), r11(2 * n), r12(2 * n), r33(2 * n), xm(2 * n), xw(2 * n);
    for (int k = 0; k < 4 * K; k++) out[k] = 0.;
    // directions in the reference loop order J = -N..-1, 1..N (SOS_SURFACE.F:1346)
    for (int q = 0; q < 2 * n; q++) {
        const int j = q < n ? -(n - q) : q - n + 1;
        xm[q] = j > 0 ? mu[j - 1] : -mu[-j - 1];
        xw[q] = j > 0 ? chr[j - 1] : chr[-j - 1];
        double c = sqrt(.5 * (1 + xm[q]));
        const double a = sqrt(ind * ind - 1.0 + c * c);
        const double b = ind * ind * c;
        const double rl = -(b - a) / (b + a);
        const double rr = (c - a) / (c + a);
        r11[q] = .5 * (rl * rl + rr * rr);
        r12[q] = .5 * (rl * rl - rr * rr);
        r33[q] = rl * rr;
    }
    double *PL = pl.data() + 1;   // PL(-1:OS_NS+1)
    for (int q = 0; q < 2 * n; q++) {            // :1387-1400
        const double x = r11[q] * xw[q], xrmu = xm[q];
        PL[-1] = 0.; PL[0] = 1.;
        for (int k = 0; k <= os_ns; k++) {
            PL[k + 1] = ((2 * k + 1.) * xrmu * PL[k] - k * PL[k - 1]) / (k + 1.);
            beta[k] = beta[k] + x * PL[k];
        }
    }
    for (int k = 0; k <= os_ns; k++) beta[k] = (2 * k + 1) * beta[k] * .5;
    for (int q = 0; q < 2 * n; q++) {            // :1433-1456
        const double xxx = xw[q] * r12[q], xx = xw[q] * r33[q], xrmu = xm[q];
        pol[0] = 0.; pol[1] = 0.;
        PL[-1] = 0.; PL[0] = 1.;
        pol[2] = 3. * (1. - xrmu * xrmu) / 2. / sqrt(6.0);
        for (int k = 2; k <= os_ns; k++) {
            const double d = (2. * k + 1.) / sqrt(1.0 * (k + 3.) * (k - 1.));
            const double e = sqrt(1.0 * (k + 2.) * (k - 2.)) / (2. * k + 1.);
            pol[k + 1] = d * (xrmu * pol[k] - e * pol[k - 1]);
            gamma[k] = gamma[k] + xxx * pol[k];
        }
        for (int k = 0; k <= os_ns; k++) {
            PL[k + 1] = ((2. * k + 1.) * xrmu * PL[k] - k * PL[k - 1]) / (k + 1.);
            delta[k] = delta[k] + xx * PL[k];
        }
    }
    for (int k = 0; k <= os_ns; k++) {
        delta[k] = delta[k] * (2. * k + 1.) * .5;
        gamma[k] = gamma[k] * (2. * k + 1.) * .5;
    }
    for (int i = 2; i <= os_ns; i++) {           // :1521-1546 ; CO1, CO2 are REAL*4 expressions
        const float co1f = 4 * (2 * i + 1.f) / (float)i / (i - 1.f) / (i + 1.f) / (i + 2.f);
        const float co2f = i * (i - 1.f) / ((i + 1.f) * (i + 2.f));
        const double co1 = co1f;
        double co2 = co2f;
        const double co3 = co2 * delta[i];
        co2 = co2 * beta[i];
        const int nn = (int)(i * .5f), mm = (int)((i - 1) * .5f);
        double som1 = 0., som2 = 0., som3 = 0., som4 = 0.;
        for (int j = 1; j <= nn; j++) {
            const double x2 = (double)((i - 1.f) * (i - 1.f) - 3.f * (2 * j - 1.f) * (i - j));
            som1 = som1 + x2 * beta[i - 2 * j];
            som2 = som2 + x2 * delta[i - 2 * j];
        }
        for (int j = 0; j <= mm; j++) {
            const double x2 = (double)((i - 1.f) * (i - 1.f) - 3.f * j * (2 * i - 2 * j - 1.f));
            som3 = som3 + x2 * beta[i - 2 * j - 1];
            som4 = som4 + x2 * delta[i - 2 * j - 1];
        }
        zeta[i] = co3 - co1 * (som2 - som3);
        alpha[i] = co2 - co1 * (som1 - som4);
    }
    for (int k = 0; k < 4 * K; k++) out[k] = through_e15_8(out[k]);
    return SOSGPU_OK;
}

extern "C" int sosgpu_glitter(int device, int n, const double *mu, const double *chr, double wind, double ind,
                              int os_nb, int os_ns, int os_nm, float *d_rsurf, int32_t *d_il, double *d_e, void *stream)
{
    if (n < 1 || n > 85 || !mu || !chr || !d_rsurf || !d_il || !d_e) return SOSGPU_E_ARG;
    if (os_nb < 0 || os_ns < 2 || os_nm < os_nb + os_ns || os_nm > 2000) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    std::vector<double> fcoef((size_t)4 * (os_ns + 1));
    int rc = sosgpu_mat_fresnel_host(n, mu, chr, ind, os_ns, fcoef.data());
    if (rc) return rc;
    const size_t cnt = (size_t)n + fcoef.size();
    TmpBuf tb(device, cnt * sizeof(double));
    if (!tb.p) return SOSGPU_E_HIP;
    double *d_buf = (double *)tb.p;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d_buf, mu, n * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_buf + n, fcoef.data(), fcoef.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        launch_glitter(n, d_buf, sigma2_of_wind(wind), os_nb, os_ns, os_nm, d_buf + n, d_il, d_e, d_rsurf, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    HIPCHK(e);
    return SOSGPU_OK;
}

static LandTerms land_terms(const sosgpu_land *land)
{
    LandTerms t;
    memset(&t, 0, sizeof t);
    if (land && land->isurf >= 3) {
        t.iroujean = 1;
        t.irondeaux = land->isurf == 4; t.ibreon = land->isurf == 5; t.imaignan = land->isurf == 7;
        t.k0 = land->k0; t.k1 = land->k1; t.k2 = land->k2; t.coef_c = land->coef_c;
    }
    return t;
}

extern "C" int sosgpu_trphi(sosgpu_ctx *cx, int nf, const double *d_rec, double tau, double tauout, int nphi,
                            const double *d_phi, int igli, double wind, const sosgpu_land *land, double *d_out, void *stream)
{
    if (!cx || nf < 1 || nf > cx->d.smax + 1 || !d_rec || nphi < 1 || !d_phi || !d_out) return SOSGPU_E_ARG;
    if (land && (land->isurf < 3 || land->isurf > 7)) return SOSGPU_E_ARG;
    if (land && land->isurf == 6) return SOSGPU_E_UNSUPPORTED;       // Nadal: refused by the reference's SOS_PROC as well
    HIPCHK(hipSetDevice(cx->device));
    launch_trphi(cx->d, nf, d_rec, tau, tauout, nphi, d_phi, igli, sigma2_of_wind(wind), cx->ind_surf, land_terms(land), d_out,
                 (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}

extern "C" int sosgpu_land_surface(int device, const sosgpu_land *land, int n, const double *mu, const double *chr, double ind,
                                   int os_nb, int os_ns, int os_nm, float *d_rsurf, int32_t *ier_out, void *stream)
{
    if (!land || land->isurf < 3 || land->isurf > 7 || n < 1 || n > 85 || !mu || !chr || !d_rsurf) return SOSGPU_E_ARG;
    if (land->isurf == 6) return SOSGPU_E_UNSUPPORTED;               // Nadal: refused by the reference's SOS_PROC as well
    if (os_nb < 0 || os_ns < 2 || os_nm < os_nb + os_ns || os_nm > 2000) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> fcoef((size_t)4 * (os_ns + 1), 0.);
    if (land->isurf > 3) {
        int rc = sosgpu_mat_fresnel_host(n, mu, chr, ind, os_ns, fcoef.data());
        if (rc) return rc;
    }
    const size_t npairs = (size_t)n * (n + 1) / 2, nn = (size_t)n * n, cnt = (size_t)(os_nb + 1) * 9 * nn;
    // one allocation: mu | fcoef | e_nn [N^2][os_nb+1] | e [npairs][os_nm+1] | il_nn, il, err (int32) | tmp matrices (float)
    const size_t nd = (size_t)n + fcoef.size() + nn * (os_nb + 1) + npairs * (os_nm + 1);
    const size_t ni = nn + npairs + 2;
    TmpBuf tb(device, nd * 8 + ni * 4 + cnt * 4 + 64);
    if (!tb.p) return SOSGPU_E_HIP;
    char *buf = (char *)tb.p;
    double *d_mu = (double *)buf, *d_fc = d_mu + n, *d_enn = d_fc + fcoef.size(), *d_e = d_enn + nn * (os_nb + 1);
    int32_t *d_ilnn = (int32_t *)(d_e + npairs * (os_nm + 1)), *d_il = d_ilnn + nn, *d_err = d_il + npairs;
    float *d_tmp = (float *)(d_err + 2);
    int32_t err[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(d_mu, mu, n * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_fc, fcoef.data(), fcoef.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, 2 * sizeof(int32_t), st);
    if (e == hipSuccess) {
        launch_land(land->isurf, n, d_mu, land->k0, land->k1, land->k2, land->coef_c, os_nb, os_ns,
                    os_nm, d_fc, d_enn, d_ilnn, d_e, d_il, d_tmp, d_rsurf, d_err, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(err, d_err, sizeof err, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    HIPCHK(e);
    if (ier_out) *ier_out = err[0] ? -1 : 0;
    return SOSGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// SOS_PROFILE on the device (profile.hip): the no-gas profile (SOS_PROFIL.F:349-489), the same for every bin of a wavelength, by
// one wavefront, the per-bin gas step by one wavefront per bin.
// ---------------------------------------------------------------------------------------------
namespace {
const int kOsNt = 600, kOsNtMin = 100;
const double kTcouche = (double)0.005f, kTFirst = (double)0.0002f;

// Level count and the two optical-depth steps of the no-gas profile (SOS_PROFIL.F:349-366): returns NT or -1 (more than
// CTE_OS_NT levels, or no scatterer at all).  The levels themselves are placed on the device (k_profile_nogas).
int profile_nogas_grid(double tr, double ta, double *t_first, double *t_layer)
{
    int nt;
    const double ttot = tr + ta;
    if ((ttot / kOsNtMin) <= kTFirst) { nt = kOsNtMin; *t_layer = ttot / nt; *t_first = *t_layer; }
    else if ((ttot / kOsNtMin) < kTcouche) { nt = kOsNtMin + 1; *t_first = kTFirst; *t_layer = (ttot - *t_first) / kOsNtMin; }
    else { *t_first = kTFirst; nt = (int)((ttot - *t_first) / kTcouche); *t_layer = (ttot - *t_first) / nt; nt = nt + 1; }
    if (nt > kOsNt || !(ttot > 0.)) return -1;
    return nt;
}
}  // namespace

extern "C" int sosgpu_profile(sosgpu_ctx *cx, int nb, double tr, double hr, double ta, double ha, int absprofil,
                              int nblev, const double *d_altabs, const double *d_tabs,
                              double a_tronc, double piz, double piztr, double zout, int lp,
                              double *d_prof, int32_t *d_nt, int32_t *d_iborm, double *d_zprof,
                              int32_t *d_jout, double *d_zz, double *d_scal, const double *d_nogas, void *stream)
{
    if (!cx || nb < 1 || lp < 2 || !d_prof || !d_nt || !d_iborm || !d_zprof || !d_scal) return SOSGPU_E_ARG;
    if ((d_jout == nullptr) != (d_zz == nullptr)) return SOSGPU_E_ARG;
    if (d_tabs && (!d_altabs || nblev < 2)) return SOSGPU_E_ARG;
    if (d_tabs && nblev > SOS_PROF_NBLEV_MAX) return SOSGPU_E_UNSUPPORTED;
    if (!(hr > 0.) || !(ha > 0.) || tr < 0. || ta < 0.) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(cx->device));
    hipStream_t st = (hipStream_t)stream;
    const int NG = SOSGPU_NOGAS_LEVELS;
    double t_first = 0., t_layer = 0.;
    const int nt_ng = profile_nogas_grid(tr, ta, &t_first, &t_layer);
    if (nt_ng < 0) return SOSGPU_E_UNSUPPORTED;        // more than CTE_OS_NT levels (IER = -1 in the reference)
    if (lp <= nt_ng) return SOSGPU_E_ARG;
    // the no-gas profile is made on `stream` in front of the bins' kernel (unless the caller queued it earlier, sosgpu_profile_nogas):
    // nothing is waited for, so profiles, solve and aggregate of a wavelength queue up behind one another without a host stall
    const double *ngp = d_nogas;
    if (!ngp) {
        if (!cx->prof_ng) { if (int rc = dev_alloc(cx, &cx->prof_ng, (size_t)4 * NG)) return rc; }
        else if (cx->prof_ng_stream != st) HIPCHK(sync_ctx_streams(cx));    // an earlier call on another stream may still be reading it
        cx->prof_ng_stream = st;
        launch_profile_nogas(tr, hr, ta, ha, nt_ng, t_first, t_layer, cx->prof_ng, NG, st);
        HIPCHK(hipGetLastError());
        ngp = cx->prof_ng;
    }
    ProfileArgs a;
    a.nb = nb; a.lp = lp; a.nblev = nblev; a.absprofil = d_tabs ? absprofil : 7; a.smax = cx->d.smax; a.nt_ng = nt_ng;
    a.tr = tr; a.hr = hr; a.ta = ta; a.ha = ha; a.a_tronc = a_tronc; a.piz = piz; a.piztr = piztr; a.zout = zout;
    a.altabs = d_altabs; a.tabs = d_tabs;
    a.z_ng = ngp; a.h_ng = ngp + NG; a.pca_ng = ngp + 2 * NG; a.pcm_ng = ngp + 3 * NG;
    a.prof = d_prof; a.zprof = d_zprof; a.zz = d_zz; a.scal = d_scal; a.nt = d_nt; a.iborm = d_iborm; a.jout = d_jout;
    launch_profile(a, st);
    HIPCHK(hipGetLastError());
    note_stream(cx, st);
    return SOSGPU_OK;
}

extern "C" int sosgpu_profile_nogas(int device, double tr, double hr, double ta, double ha, double *d_nogas, void *stream)
{
    if (!d_nogas || !(hr > 0.) || !(ha > 0.) || tr < 0. || ta < 0.) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    double t_first = 0., t_layer = 0.;
    const int nt_ng = profile_nogas_grid(tr, ta, &t_first, &t_layer);
    if (nt_ng < 0) return SOSGPU_E_UNSUPPORTED;
    HIPCHK(hipSetDevice(device));
    launch_profile_nogas(tr, hr, ta, ha, nt_ng, t_first, t_layer, d_nogas, SOSGPU_NOGAS_LEVELS, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}

extern "C" int sosgpu_absprofile(int device, int nb, int nlev, int nterm, const int32_t *d_ik, const double *d_xk,
                                 const double *d_ro, double *d_tabs, void *stream)
{
    if (nb < 1 || nlev < 2 || nlev > SOS_PROF_NBLEV_MAX || nterm < 1 || !d_ik || !d_xk || !d_ro || !d_tabs) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    launch_absprofile(nb, nlev, nterm, d_ik, d_xk, d_ro, d_tabs, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}

extern "C" int sosgpu_mie(int device, int nbmu, const double *xmu, double rn, double in, int nalpha, const double *alphas,
                          float *d_rec, double *d_g, void *stream)
{
    if (nbmu < 1 || nbmu > 100 || !xmu || nalpha < 1 || !alphas || !d_rec || !d_g) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const int W = 2 * nbmu + 1;
    // ascending size parameters (SOS_MIE's grid): those whose 11 arrays of 2 alpha + 24 terms fit LDS form a prefix
    const double lds_alpha = 850.;
    double amax = 0., alds = 0.;
    int n_lds = 0;
    for (int i = 0; i < nalpha; i++) {
        if (!(alphas[i] > 0.) || (i && alphas[i] < alphas[i - 1])) return SOSGPU_E_ARG;
        amax = alphas[i];
        if (alphas[i] <= lds_alpha) { n_lds = i + 1; alds = alphas[i]; }
    }
    if (2 * amax + 24 > 10000) return SOSGPU_E_UNSUPPORTED;                      // CTE_MIE_DIM, SOS.h:117
    const size_t nscr = n_lds < nalpha ? mie_scratch_doubles(amax, nalpha - n_lds) : 0;
    TmpBuf tb(device, (size_t)(W + nalpha + nscr) * sizeof(double) + 64);
    if (!tb.p) return SOSGPU_E_HIP;
    char *buf = (char *)tb.p;
    double *d_xmu = (double *)buf, *d_al = d_xmu + W, *d_scr = d_al + nalpha;
    int32_t *d_err = (int32_t *)(d_scr + nscr);
    int32_t err = 0;
    hipError_t e = hipMemcpyAsync(d_xmu, xmu, W * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_al, alphas, nalpha * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, sizeof(int32_t), st);
    int rc = 0;
    if (e == hipSuccess) {
        rc = launch_mie(nalpha, nbmu, d_xmu, rn, in, d_al, n_lds, alds, amax, nscr ? d_scr : nullptr, d_rec, d_g, d_err, st);
        if (rc == 0) e = hipGetLastError();
    }
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(&err, d_err, sizeof err, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    if (rc == -3) return SOSGPU_E_UNSUPPORTED;
    if (rc == -2) { g_last_hip = (int)hipGetLastError(); return SOSGPU_E_HIP; }
    HIPCHK(e);
    return err ? SOSGPU_E_UNSUPPORTED : SOSGPU_OK;
}

extern "C" int sosgpu_granu(int device, int nbmu, int nalpha, const float *d_rec, int igranu, double v1, double v2, double v3,
                            double wa, double alphaf, double *out, void *stream)
{
    if (nbmu < 1 || nbmu > 100 || nalpha < 1 || !d_rec || igranu < 1 || igranu > 2 || !out || !(wa > 0.)) return SOSGPU_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nout = (size_t)3 + 3 * (2 * nbmu + 1), nwork = (size_t)3 * nalpha + 1;
    TmpBuf tb(device, (nout + nwork) * sizeof(double));
    if (!tb.p) return SOSGPU_E_HIP;
    double *d_out = (double *)tb.p, *d_work = d_out + nout;
    launch_granu(nalpha, nbmu, d_rec, igranu, v1, v2, v3, wa, alphaf, d_work, d_out, st);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, nout * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    HIPCHK(e);
    return SOSGPU_OK;
}

extern "C" int sosgpu_granu_batch(int device, int nbmu, int count, const sosgpu_granu_job *jobs, double *d_out, double *d_work,
                                  size_t work_stride, void *stream)
{
    if (nbmu < 1 || nbmu > 100 || count < 0 || (count && (!jobs || !d_out || !d_work))) return SOSGPU_E_ARG;
    for (int k = 0; k < count; k++) {
        const sosgpu_granu_job &j = jobs[k];
        if (j.nalpha < 1 || !j.d_rec || j.igranu < 1 || j.igranu > 2 || !(j.wa > 0.) || work_stride < (size_t)3 * j.nalpha + 1)
            return SOSGPU_E_ARG;
    }
    if (!count) return SOSGPU_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SOSGPU_E_NODEVICE;
    if (device < 0 || device >= ndev) return SOSGPU_E_ARG;
    HIPCHK(hipSetDevice(device));
    launch_granu_batch(count, nbmu, jobs, d_work, work_stride, d_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return SOSGPU_OK;
}

// Diagnostic: the streamed solver's scratch of this context (device pointer, size in doubles) and, after an order-parallel
// solve, where its I3 hand-over block starts (doubles from the start; 0 when the last solve did not use that form).
extern "C" int sosgpu_debug_scratch(sosgpu_ctx *cx, double **d_scratch, size_t *doubles, size_t *spec_i3_offset)
{
    if (!cx || !d_scratch || !doubles || !spec_i3_offset) return SOSGPU_E_ARG;
    *d_scratch = cx->scratch;
    *doubles = cx->scratch_doubles;
    *spec_i3_offset = cx->dbg_spec_i3;
    return SOSGPU_OK;
}

// Diagnostic: per-bin phase cycle counters [nb][8] (filled only by builds with -DSOS_PROFILE_PHASES).
extern "C" int sosgpu_debug_phase_buffer(sosgpu_ctx *cx, unsigned long long *d_phase)
{
    if (!cx) return SOSGPU_E_ARG;
    cx->phase = d_phase;
    return SOSGPU_OK;
}
