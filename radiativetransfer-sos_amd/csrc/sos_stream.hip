// csrc/sos_stream.hip -- successive-orders solver for bins whose level grid does not fit LDS (gfx950).
//
// Reference profiles have NT = 100 ... 600 levels (CTE_OS_NT_MIN / CTE_OS_NT, SOS.h:229,202): the radiance field of one
// bin, 6N rows x (NT+1) levels x 8 B, is 0.2 ... 1.2 MB at N = 41 and cannot live in the 160 KB of LDS the way the field of
// the 30-layer benchmark bins does (sos_os.hip).  Same routine chain as sos_os.hip -- SOS_OS (src/SOS_OS.F:303-1674) with
// SOS_FSOURCE_ORDRE1 (:2431), SOS_FSOURCE_ORDREIG (:2663), SOS_INTEGR_EPOPT (:2222), SOS_FSOURCE_DIFF_FRESNEL1 (:3106), the
// stop tests and SOS_AJOUT_QUEUE -- with the field kept in a per-bin scratch in HBM/L2 and STREAMED through LDS in chunks
// of 32 levels, ONE pass from the top of the atmosphere to the ground per scattering order:
//
//   per chunk (levels l0 .. L):   LDS-DMA (global_load_lds, 16 B/lane) of the chunk [Q+ | X-] of order ig-1
//                                 -> X+ = Q+ + P Xin (see below) -> MFMA contraction S = M X (gemm_source, exactly the
//                                 LDS-resident kernel's) -> formal solution of the chunk in LDS -> store [Q+ | X-] of order ig.
//
// The source is local in the level (S(:, i) = M X(:, i)); only the formal solution couples levels.  Down-going rows are
// swept exactly, the running value crossing chunk borders in a register.  Up-going rows would need a second pass from the
// ground upwards; instead the formal solution is split per chunk into its particular and homogeneous parts (a blocked scan):
//   X+_i = Q_i + P_i Xin(c),   Q_i = solution of the chunk with zero inflow at its bottom level L,
//   P_i = prod_{k=i}^{L-1} exp(-dtau_k/mu)  (attenuation from L up to i),   Xin(c) = X+_L,
// with the chunk-to-chunk recurrence Xin(c-1) = A(c) + B(c) Xin(c) (A: one more layer step on Q at the top level of chunk
// c, B = exp(-dtau_{l0-1}/mu) P_{l0}) resolved after the last chunk, when the ground value (the reflection of the previous
// order, SOS_OS.F:1166-1239) enters.  Q is what the scratch keeps; the homogeneous part is added when the chunk is next
// loaded.  Algebraically the reference recurrence X_i = t X_{i+1} + p S_i + w S_{i+1}; rounding differs at 1e-16 of the field.
// Scratch traffic per scattering order: the field once in and once out (the field-in-HBM variant of round 1 moved it three
// times, with the B operands and both sweeps latency-bound on global memory).
//
// Launch forms (api.hip, os_solve_impl): one workgroup per bin running all its Fourier orders (large batches); the
// ORDER-PARALLEL form for few bins -- one workgroup per (bin, Fourier order), the stop tests replayed by k_sos_stream_replay
// (SosBins::spec_k); a persistent launch taking (order, bin) tasks from per-XCD queues (PERSIST, opt-in); order-synchronous
// launches (SosBins::s_begin / s_end, opt-in).  All of them run the same task body, run_task.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "sos_dev.h"
#include "kernels.h"

// SOS_MULTI (set by sos_stream_multi.hip, which includes this file): the same kernels under other names, with a per-bin
// wavelength context read from a device table (see sos_os.hip)
#ifdef SOS_MULTI
#define k_sos_stream k_sos_stream_multi
#define launch_sos_stream launch_sos_stream_multi
#endif

#ifndef SOS_STREAM_NT
#define SOS_STREAM_NT 2
#endif
namespace {
#ifndef SOS_STREAM_TOP_BARRIER
#define SOS_STREAM_TOP_BARRIER 0   // 1: workgroup barrier between the store pass of a chunk and the staging of the next one
#endif
#ifndef SOS_STREAM_PREFETCH
#define SOS_STREAM_PREFETCH 0      // 1: request chunk c+1 from inside the store pass of chunk c.  Measured 3.4 % SLOWER (21.4 k vs
                                   // 22.1 k bins/s, profiles/r02_stream_experiments.txt): the per-unit address arithmetic of the
                                   // requests costs more vector issue than the hidden latency returns.  Kept as an experiment switch.
#endif
#ifndef SOS_STREAM_COLS
#define SOS_STREAM_COLS 32
#endif
constexpr int COLS = SOS_STREAM_COLS;      // levels per chunk (two 16-column MFMA tiles; 16 = experiment: three workgroups per CU)
constexpr int VPAD = 8;       // level vectors are stored with a +1 offset (entry e = level e-1) and a few spare entries
}

// Row capacity of the field layout.  Round 2 sized it by the wave shape alone, KHM = 16 NW RTWH rows per half: 64, 128, 256.  A half
// system of 66 ... 96 rows (N = 22 ... 32 -- the reference's default of 24 Gauss angles is N = 25) then moved 128-row chunks through
// LDS and HBM: N = 22 ran 32 % slower than N = 21 for 5 % more work (profiles/r03_n_sweep.txt).  The layout is now sized in row
// tiles, KHT = 5 or 6 for those direction counts (KHM = 80 / 96, FS = 162 / 194: 41 / 50 KB chunks instead of 66 KB), the wave
// shape staying <4,2> (wave 0, or waves 0 and 1, carry a second tile).  KHM = 16 KHT, FS = 2 KHM + 2 (still = 2 mod 4 doubles:
// conflict-free ds_read_b128 of the B operands), NS = directions capacity.
__host__ __device__ constexpr int skhm(int kht) { return 16 * kht; }
__host__ __device__ constexpr int sfs(int kht) { return 2 * skhm(kht) + 2; }
__host__ __device__ constexpr int sns(int kht) { return (skhm(kht) / 3 + 1) & ~1; }

// scratch of one bin, in doubles (lpb = level capacity, a multiple of COLS)
__host__ __device__ inline size_t stream_scratch_doubles(int nw, int kht, int lpb)
{
    const size_t fs = sfs(kht), ns = sns(kht), khm = skhm(kht), nch = lpb / COLS;
    const size_t d = (size_t)lpb * fs + (size_t)(lpb + 1) * ns + 7 * (size_t)(lpb + VPAD) + nch * (2 * khm + 2 * ns) +
                     2 * 64 * (size_t)nw + 8;                  // + state carried from one launch to the next (i4, i5 per thread)
    return (d + 15) & ~(size_t)15;
}

// Homogeneous part of the up-going field of U levels: P <- P (1 - a), X = Q + P xin (see the header); q, qa move upwards.
// (bases moved to the low end of the block and pinned: non-negative immediate offsets, see scan_block)
// KHM > 0: the closed level is stored in parity form at once, X^A = X+ + X- in its own slot and X^B = X+ - X- in the slot of
// the down-going row with the same index (KHM doubles further): the contraction that follows reads its B operands without
// forming the sums (SOS_PRECOMBINE_STREAM), and no extra pass or barrier is needed -- every (row, level) of the chunk is closed here
// by exactly one thread.
template <int U, int FS, int NS, int KHM>
__device__ __forceinline__ void fix_block(lds_f64 *&q, const lds_f64 *&qa, double &P, double xi)
{
    q -= U * FS; qa -= (U - 1) * NS;
    lds_pin(q); lds_pin(qa);
    double av[U], qv[U], xm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        av[u] = qa[(U - 1 - u) * NS]; qv[u] = q[(U - 1 - u) * FS];
        if (KHM) xm[u] = q[(U - 1 - u) * FS + KHM];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { P = P - P * av[u]; qv[u] = qv[u] + P * xi; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (KHM) { q[(U - 1 - u) * FS] = qv[u] + xm[u]; q[(U - 1 - u) * FS + KHM] = qv[u] - xm[u]; }
        else q[(U - 1 - u) * FS] = qv[u];
    }
    qa -= NS;
}

// linear LDS-DMA copy of `units` 16-byte units global -> LDS (wave-uniform LDS base + lane * 16, exec-masked tail)
// AUX = 2: non-temporal (the field of a bin is read once per scattering order: it must not push the source operator, which
// every workgroup of the XCD re-reads, out of L2)
template <int NTH, int AUX>
__device__ __forceinline__ void glds_copy(const double *g, double *l, int units, int t)
{
    // Wave-uniform trip count and a wave-uniform running base on both sides: the global address is (scalar base + 16 t) and the
    // LDS base (M0) advances by a constant, so a unit costs scalar instructions only -- no per-lane address arithmetic, no
    // per-lane loop predicate (vector issue is the scarce resource of this kernel, see sos_dev.h).
    const char *gb = reinterpret_cast<const char *>(g);
    // LDS base of this wave as a scalar (M0 is written from it, advanced by scalar adds)
    const unsigned lb = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(l) + (size_t)(t & ~63) * 16));
    unsigned toff = (unsigned)t * 16u;
    const int full = units / NTH, rem = units - full * NTH;
#pragma unroll 1
    for (int i = 0; i < full; i++) {
        asm volatile("" : "+v"(toff));      // keep (scalar base + 32-bit lane offset) apart: the scalar-base addressing form
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb + (size_t)i * (NTH * 16) + toff),
                                         (__attribute__((address_space(3))) void *)(unsigned long)(lb + (unsigned)i * (NTH * 16)), 16, 0, AUX);
    }
    if (t < rem)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb + (size_t)full * (NTH * 16) + toff),
                                         (__attribute__((address_space(3))) void *)(unsigned long)(lb + (unsigned)full * (NTH * 16)), 16, 0, AUX);
}

// Workgroups per CU.  The four-tile layout (N <= 21: 42 KB of LDS per workgroup) runs THREE (round 3): 168 instead of 256
// registers per lane, and the third wave per SIMD hides more latency than the spills cost -- N = 13 / 17 / 21 at NT = 120:
// 103 / 93 / 81 -> 114 / 103 / 87 k bins/s -- and so does the five-tile layout (N <= 26: 52 KB): N = 25 60.1 -> 65.4 k bins/s
// (profiles/r03_n_sweep.txt).  SOS_STREAM_3WG_KHT: the widest layout that does.
#ifndef SOS_STREAM_3WG_KHT
#define SOS_STREAM_3WG_KHT 5
#endif
#define SOS_STREAM_MIN_WG(NW, KHT) ((NW) == 4 ? ((COLS == 16 || (KHT) <= SOS_STREAM_3WG_KHT) ? 3 : 2) : 1)
template <int NW, int RTWH, bool ZO, bool SURF, bool PERSIST, int KHT = NW * RTWH>
__global__ __launch_bounds__(64 * NW, SOS_STREAM_MIN_WG(NW, KHT)) void k_sos_stream(const SosDev cx_arg, const SosBins bn)
{
    SOS_BIND_CTX(cx, cx_arg, bn);
    const auto &cx_kernel = cx;
    const SosBins &bn_kernel = bn;
    extern __shared__ double smem[];
    constexpr int CT = COLS / 16, NTH = 64 * NW, HW = NW / 2;
    static_assert(KHT > NW * (RTWH - 1) && KHT <= NW * RTWH, "layout rows: more than RTWH - 1 tiles per wave, at most RTWH");
    constexpr int KHM = skhm(KHT), FS = sfs(KHT), NS = sns(KHT);
    constexpr int VL = COLS + VPAD;        // chunk copy of a level vector: entry e = level (layer) l0 - 1 + e
    const int N = cx.n, KP = cx.kp, KH = cx.kh, W = cx.w;
    const int LPB = bn.lpb, NCH = LPB / COLS, VS = LPB + VPAD;
    double *cbuf = smem;                   // [COLS][FS]  the chunk: field of order ig-1, then source, then field of order ig
    double *gnd = cbuf + COLS * FS;        // down-going field at the ground, order ig-1: [3][NS], or -- SURF -- [KHM] in
                                           // half-system order (B operand of ground_mfma), as in sos_os.hip
    double *red = gnd + 3 * NS + 2;        // [16]
    double *i3s = red + 16;                // [2][NS] I3 of the I rows (flux integrals)
    double *lga = i3s + 2 * NS;            // [NS] Gauss weights, [NS] mu
    double *lmu = lga + NS;
    double *bcv = i3s;                     // SURF: [KHM] result of ground_mfma, over i3s | lga | lmu
    double *catt = lmu + NS;               // [COLS][NS] 1 - exp(-dtau/mu): row r = layer l0 - 1 + r
    double *cvec = catt + COLS * NS;       // [7][VL] level vectors of the chunk
    double *cidt = cvec, *cxdel = cvec + VL, *cydel = cvec + 2 * VL, *ccxd = cvec + 3 * VL;

    // One TASK = the Fourier orders [s_begin, s_end) of bin b.  Returns true when the Fourier series of the bin has ended
    // (stop test, IBORM reached, malformed bin).  Plain launches run one task per workgroup (b = blockIdx.x); the persistent
    // form (PERSIST) takes tasks from per-XCD queues, see below.
    // (The per-thread constants are formed inside, from an opaque copy of the thread index: hoisted out of the persistent
    // form's task loop they would stay live across the whole solve -- 60 more spilled registers.)
    // PERSIST: the two kernel arguments are re-bound per task through an opaque copy of the kernarg pointer -- as plain
    // by-value arguments every field the task reads would be loaded once in front of the task loop and kept (or spilled) for
    // the whole solve.
    auto bind_cx = [&]() -> decltype(auto) {
        if constexpr (PERSIST) {
            const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            return (*(const SosDevK *)ka);
        } else return (cx_kernel);
    };
    auto bind_bn = [&]() -> decltype(auto) {
        if constexpr (PERSIST) {
            const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            return (*(const SosBinsK *)(ka + SOS_KERNARG_BINS_OFFSET));
        } else return (bn_kernel);
    };
    // rc / rw: scratch regions holding the bin's constants (attenuations, level vectors, link factors, state) and this task's
    // work arrays (field, chunk inflows); the same region except in the order-parallel form (SPEC, below), where `spec` is set:
    // the task then runs ONE order without the bin's Fourier sums -- it leaves its I3 terms in bn.spec_i3 and the stop test
    // to k_sos_stream_replay.
    auto run_task = [&](const int b, const int s_begin, const int s_end, const size_t rc, const size_t rw, const bool spec) -> bool {
    decltype(auto) cx = bind_cx();
    decltype(auto) bn = bind_bn();
    int t = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t));
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool up = wv < HW;               // wave-uniform
    const int kk0 = up ? t : t - 64 * HW;
    const bool active = kk0 < 3 * N;
    const int kk = active ? kk0 : 0;
    const int cj = cx.rowmap[kk];
    const int c = cj / N;
    const int jj = cj % N;
    const int d = up ? jj : N + jj;
    const int rl = up ? kk : KHM + kk;     // field row of this thread
    const int rsv = c * 2 * N + d;
    const double mu = cx.mu[jj];
    const int recoff = c * W + N + (up ? (jj + 1) : -(jj + 1));
    const size_t mper = (size_t)2 * cx.rtph * cx.ks2h * 128;
    const int S1 = cx.smax + 1;
    const bool tile_a = wv * 16 < KH;
    const bool tile_b = RTWH > 1 && (wv + NW) * 16 < KH;
    const double *bx = cbuf + (lane & 15) * FS + 2 * (lane >> 4);
    double *pcb = cbuf + (lane & 15) * FS + (cx.prow >= 0 ? cx.prow : 0) + (lane >> 4);

    double *cbase = bn.scratch + rc * bn.scr_stride, *wbase = bn.scratch + rw * bn.scr_stride;
    double *fld = wbase;                                  // [LPB][FS]
    double *att = cbase + (size_t)LPB * FS;               // [LPB+1][NS]: row i+1 = layer i (levels i..i+1), row 0 zero
    double *vec = att + (size_t)(LPB + 1) * NS;           // [7][VS]: idtau (layer) | xdel | ydel | cxd | cyd | fxd | fyd
    const size_t off_xin = (size_t)LPB * FS + (size_t)(LPB + 1) * NS + (size_t)7 * VS;
    double *xin = wbase + off_xin;                        // [NCH][KHM]
    double *acf = xin + (size_t)NCH * KHM;                // [NCH][KHM]
    double *bcf = cbase + off_xin + (size_t)2 * NCH * KHM;   // [NCH][NS]
    double *pmid = bcf + (size_t)NCH * NS;                // [NCH][NS] attenuation from the bottom level of a chunk to its middle
    double *state = pmid + (size_t)NCH * NS;              // [8 + 2 NTH]: status | has_aer | nord | ... | i4, i5 per thread
    const int nt = uniform_i32(bn.nt[b]);
    const int iborm = uniform_i32(bn.iborm[b]);
    const int jout = ZO ? uniform_i32(bn.jout ? bn.jout[b] : 0) : 0;
    const double zz = ZO ? uniform_f64(jout ? bn.zz[b] : 0.) : 0.;
    const int jlo = (ZO && jout) ? jout - 1 : 0, jhi = (ZO && jout) ? jout : 0;
    if (nt < 1 || nt >= LPB || nt >= bn.lp || iborm < 0 || iborm > cx.smax || jout < 0 || jout > nt) {
        if (t == 0) { bn.norders[b] = -1; bn.flux[2 * b] = 0.; bn.flux[2 * b + 1] = 0.; }
        for (int i = t; i < cx.smax + 1; i += NTH) bn.iglast[(size_t)b * (cx.smax + 1) + i] = 0;
        return true;
    }
    const int nchunk = (nt + COLS) / COLS;
    const double *pf = bn.prof + (size_t)b * 3 * bn.lp;
    double *recb = bn.rec + (size_t)b * S1 * 3 * W;
    // Order-synchronous launches: a launch runs the Fourier orders [s_begin, s_end) of every bin, so that all workgroups of
    // an XCD stream the SAME source operator (240 KB at N = 41) from L2 instead of ~20 different ones from the fabric.  The
    // scratch keeps the bin constants (level vectors, attenuations, link factors) and the little state a bin carries from
    // order to order (running Fourier sums I4, I5 of every row, SOS_OS.F:1460-1473) between launches.
    const bool first = s_begin == 0 && !spec;
    if (spec && s_begin > iborm) return false;                 // an order the bin never runs
    // (state written by another workgroup -- an earlier launch, or an earlier task of a persistent launch -- is read with
    //  agent-scope loads, never through the scalar cache)
    if (!first && uniform_f64(__hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0.)
        return true;                                           // the Fourier series of this bin has already stopped
    const double htot = uniform_f64(pf[nt]), h0 = uniform_f64(pf[0]);
    const double hlo = uniform_f64(pf[jlo]), hhi = uniform_f64(pf[jhi]);
    int has_aer;

    // ---- per-bin set-up: level vectors, attenuation table, chunk link factors -> scratch -------------------------------
    __syncthreads();
  if (first) {
    if (!SURF && t < N) { lga[t] = cx.ga[t]; lmu[t] = cx.mu[t]; }
    if (SURF) for (int i = t; i < 3 * NS + 2; i += NTH) gnd[i] = 0.;
    for (int i = t; i < COLS * FS; i += NTH) cbuf[i] = 0.;
    double *v_idt = vec, *v_xd = vec + VS, *v_yd = vec + 2 * VS, *v_cxd = vec + 3 * VS, *v_cyd = vec + 4 * VS,
           *v_fxd = vec + 5 * VS, *v_fyd = vec + 6 * VS;
    double *hh = v_fyd, *dtau = v_fxd;         // set-up only: these two slots end up holding the Fresnel factors
    for (int e = t; e < VS; e += NTH) {        // entry e = level e - 1
        const int i = e - 1;
        const bool in = i >= 0 && i <= nt;
        hh[e] = in ? pf[i] : 0.;
        v_xd[e] = in ? pf[bn.lp + i] : 0.;
        v_yd[e] = in ? pf[2 * bn.lp + i] : 0.;
    }
    for (int i = t; i < NS; i += NTH) att[i] = 0.;
    // levels above nt of the last chunk stay zero in the scratch for the whole solve (they are loaded, never stored)
    for (size_t i = (size_t)(nt + 1) * FS + t; i < (size_t)nchunk * COLS * FS; i += NTH) fld[i] = 0.;
    __syncthreads();
    int aer_l = 0;
    for (int e = t; e < VS; e += NTH) {
        const int i = e - 1;
        const double dt = (i >= 0 && i < nt) ? hh[e + 1] - hh[e] : 0.;
        dtau[e] = dt;
        v_idt[e] = (i >= 0 && i < nt) ? 1.0 / dt : 0.;
        const double ch = (i >= 0 && i <= nt) ? exp(-hh[e] / cx.mus) / 4. : 0.;                      // SOS_OS.F:837-839
        v_cxd[e] = ch * v_xd[e]; v_cyd[e] = ch * v_yd[e];
        if (i >= 0 && i <= nt && v_xd[e] != 0.) aer_l = 1;
    }
    has_aer = uniform_i32(__syncthreads_or(aer_l));
    if (t == 0) { state[0] = 0.; state[1] = has_aer; }
    for (int i = t; i < nt * N; i += NTH)
        att[(size_t)(i / N + 1) * NS + i % N] = 1.0 - exp(-dtau[i / N + 1] / cx.mu[i % N]);          // SOS_OS.F:2291,2335
    __syncthreads();
    for (int e = t; e < VS; e += NTH) {        // own elements only: h_i -> fco_i XDEL_i | fco_i YDEL_i
        const int i = e - 1;
        const double fco = (i >= 0 && i <= nt) ? (exp(-2. * htot / cx.mus) / 4.) * exp(hh[e] / cx.mus) : 0.;   // SOS_OS.F:3219,3278
        v_fxd[e] = fco * v_xd[e]; v_fyd[e] = fco * v_yd[e];
    }
    for (int i = t; i < nchunk * N; i += NTH) {
        // chunk cq, direction j: attenuation from the chunk's bottom level up to its top level, times the layer above it
        const int cq = i / N, j = i % N;
        const int l0 = cq * COLS, L = min(l0 + COLS - 1, nt);
        const int mid = l0 + (L - l0) / 2;           // the up-going waves fix levels L .. mid+1, the down-going waves mid .. l0
        double P = 1.0;
        for (int k = L - 1; k >= l0; --k) {
            P = P - P * att[(size_t)(k + 1) * NS + j];
            if (k == mid + 1) pmid[cq * NS + j] = P;  // attenuation from L up to level mid + 1
        }
        if (mid + 1 >= L) pmid[cq * NS + j] = 1.0;
        if (cq > 0) P = P - P * att[(size_t)l0 * NS + j];                                           // layer l0 - 1
        bcf[cq * NS + j] = P;
    }
    __syncthreads();
  } else {
    if (!SURF && t < N) { lga[t] = cx.ga[t]; lmu[t] = cx.mu[t]; }
    if (SURF) for (int i = t; i < 3 * NS + 2; i += NTH) gnd[i] = 0.;
    for (int i = t; i < COLS * FS; i += NTH) cbuf[i] = 0.;
    has_aer = uniform_i32((int)__hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // a work region of its own: the levels above nt of the last chunk are loaded, never stored -- zero them here
    if (spec) for (size_t i = (size_t)(nt + 1) * FS + t; i < (size_t)nchunk * COLS * FS; i += NTH) fld[i] = 0.;
    __syncthreads();
  }
    const double e_sun = uniform_f64(exp(-htot / cx.mus));
    // ZO: attenuation from the bottom of its chunk up to the two output levels (up-going rows)
    double plo = 1., phi = 1.;
    if (ZO && jout && active && up) {
        const int Llo = min((jlo / COLS) * COLS + COLS - 1, nt), Lhi = min((jhi / COLS) * COLS + COLS - 1, nt);
        for (int k = Llo - 1; k >= jlo; --k) plo = plo - plo * att[(size_t)(k + 1) * NS + jj];
        for (int k = Lhi - 1; k >= jhi; --k) phi = phi - phi * att[(size_t)(k + 1) * NS + jj];
    }
    __syncthreads();

    const double usign = (c == 2 && !up) ? -1. : 1.;     // U(-mu) is stored negated (sos_dev.h, gemm_source)
    double xb = 0., xlo = 0., xhi = 0.;
    Order1 o1 = {0., 0., 0., 0., false};

    double i4 = 0., i5 = 0.;
    double sign = -1.;
    int red_slot = 0;
    int nord = 0;
    bool finished = false;
    if (!first) {
        i4 = state[8 + 2 * t]; i5 = state[8 + 2 * t + 1];
        nord = s_begin;
        if (s_begin & 1) sign = 1.;                   // sign = (-1)^s after the flip at the top of the loop
    }
#ifdef SOS_PROFILE_PHASES
    unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 stage wait, 1 fix-up, 2 gemm, 3 write-back, 4 sweeps, 5 store, 6 pass end + tests, 7 order-1 passes
#endif
    PH_T0();
    const int s_last = min(iborm, s_end - 1);
    if (s_begin > iborm) finished = true;
    for (int s = s_begin; s <= s_last; ++s) {  // SOS_OS.F:872
        sign = -sign;
        // ground reflection of the down-going field of the previous order (SOS_OS.F:1166-1239); called by every thread
        const double *gop = SURF ? cx.mp_gnd + (size_t)s * cx.rtph * cx.ks2h * 128 : nullptr;
        const int gstep = (jj < cx.nwgt || !SURF) ? cx.nwgt : N - cx.nwgt;
        auto ground_bc = [&]() -> double {
            double v = 0.;
            if (SURF) {
                if (tile_b) ground_mfma<RTWH, NW>(gop, cx.ks2h, gnd, bcv, lane, wv);
                else if (tile_a) ground_mfma<1, NW>(gop, cx.ks2h, gnd, bcv, lane, wv);
                __syncthreads();
                if (!(active && up)) return 0.;
                v = bcv[kk];
            } else {
                if (!(active && up)) return 0.;
                if (c == 0 && cx.ro != 0. && s == 0) {
                    double lsol = 0.;
#pragma unroll 1
                    for (int j = 0; j < N; j++) lsol = lsol + lga[j] * gnd[j] * lmu[j];
                    v = 2 * lsol * cx.ro;
                }
            }
            if (cx.ifresnel == 1) {
                unsigned jf8 = (unsigned)jj * 8u;
                const double f11 = *lane_ptr(cx.fres, jf8), f12 = *lane_ptr(cx.fres + N, jf8), f33 = *lane_ptr(cx.fres + 2 * N, jf8);
                const double *g0 = SURF ? gnd + kk - c * gstep : gnd + jj;
                const int gs = SURF ? gstep : NS;
                if (c == 0) v = v + f11 * g0[0] + f12 * g0[gs];
                else if (c == 1) v = v + f12 * g0[0] + f11 * g0[gs];
                else v = v + f33 * g0[2 * gs];
            }
            return v;
        };

        const double *svp = cx.sv + (size_t)s * 4 * KP;
        if (active) {
            o1.sva = svp[rsv] * usign; o1.svr = svp[KP + rsv] * usign;
            o1.fres = cx.ifresnel == 1;
            if (o1.fres) { o1.sfa = svp[2 * KP + rsv] * usign; o1.sfr = svp[3 * KP + rsv] * usign; }
        }
        double bc = 0., dirterm = 0.;
        if (active && up) {                                                             // SOS_OS.F:970-992
            double xr = 0.;
            if (c == 0 && cx.ro != 0. && s == 0) { bc = cx.ro * cx.mus * e_sun; xr = bc; }
            if (SURF) {
                bc = bc + cx.rdir[((size_t)s * 3 + c) * N + jj] * (e_sun / mu);
                dirterm = bc - xr;                                                       // SOS_OS.F:1070-1072
            }
        }
        const double *mpa = cx.mp_aer + (size_t)s * mper;
        const double *vtp = cx.mp_vt + (size_t)(s <= 2 ? s : 0) * cx.ks2h * 128;
        const double *ufp = cx.mp_uf + (size_t)(s <= 2 ? s : 0) * cx.rtph * 64;
        const bool fold = s <= 2 && has_aer && cx.prow >= 0;

        // per-lane byte offsets of the scalar-base accesses (lane_ptr); level-vector staging: entry e of the 7 x VL chunk copy
        // comes from vector e / VL at level l0 - 1 + e % VL -- the split is per-thread constant, only l0 moves
        unsigned kk8 = (unsigned)kk * 8u, jj8 = (unsigned)jj * 8u;
        unsigned vso0 = (unsigned)(((t / VL) * VS + t % VL) * 8), vso1 = (unsigned)((((t + NTH) / VL) * VS + (t + NTH) % VL) * 8);
        auto stage_vec = [&](int lv) {
            static_assert(7 * VL <= 2 * NTH, "two entries per thread at most");
            if (t < 7 * VL) cvec[t] = *lane_ptr(vec + lv, vso0);
            if (t + NTH < 7 * VL) cvec[t + NTH] = *lane_ptr(vec + lv, vso1);
        };

        // ---- one pass over the chunks = one scattering order (O1: the single-scattering source, formed on the fly) ----
        auto pass = [&](auto o1_tag, double bcv) {
            constexpr bool O1 = decltype(o1_tag)::value;
            double dn_z = 0., dn_s = 0., up_sprev = 0., q_top = 0., qlo = 0., qhi = 0.;
#pragma unroll 1
            for (int chk = 0; chk < nchunk; chk++) {
                const int l0 = chk * COLS, L = min(l0 + COLS - 1, nt), nlev = L - l0 + 1;
                // stage the chunk: field of order ig-1 (32 levels; the scratch is zero beyond nt), attenuations of layers
                // l0-1 .. l0+30, level vectors of levels l0-1 .. l0+38.  In the passes of order >= 2 every chunk but the first
                // was already requested by the store pass of the chunk before it (below).
                if (O1 || !SOS_STREAM_PREFETCH || chk == 0) {
                    // No barrier here: the only readers of the chunk buffer after the barrier that precedes the store pass are
                    // the store pass's own reads, and a 16-byte unit is stored and re-requested by the SAME thread (identical
                    // unit mapping in glds_copy and in the store loop); the attenuation / level-vector copies have no reader
                    // left.  The barrier after the staging wait below is the one every wave passes before anything is used.
                    if (SOS_STREAM_TOP_BARRIER) __syncthreads();
                    if (!O1) glds_copy<NTH, SOS_STREAM_NT>(fld + (size_t)l0 * FS, cbuf, COLS * FS / 2, t);
                    glds_copy<NTH, 0>(att + (size_t)l0 * NS, catt, COLS * NS / 2, t);
                    stage_vec(l0);
                }
                double xi = 0., pm = 1.;
                if (!O1 && active) { xi = *lane_ptr(xin + chk * KHM, kk8); if (!up) pm = *lane_ptr(pmid + chk * NS, jj8); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (!O1) PH(0);
                if (!O1) {
                    // homogeneous part of the up-going rows: X+ = Q + P Xin, P running from the bottom level upwards.  Shared
                    // by all four waves: the thread of up-going row kk does levels L .. mid+1, the thread of the down-going
                    // row with the same kk does levels mid .. l0 of that UP-going row, starting from the tabulated P(mid+1)
                    if (active) {
                        const int mid = l0 + (L - l0) / 2;
                        const int lev = up ? L : mid + 1;              // level this thread starts from (already final there)
                        lds_f64 *q = (lds_f64 *)(cbuf + (size_t)(lev - l0) * FS + kk);          // row kk of the up-going half
                        const lds_f64 *qa = (const lds_f64 *)(catt + (size_t)(lev - l0) * NS + jj);   // layer lev-1
                        double P = up ? 1.0 : pm;                       // attenuation from L up to `lev`
                        int cnt = up ? L - mid - 1 : (L > l0 ? mid - l0 + 1 : 0);
                        if (cnt < 0) cnt = 0;
                        if (up) {                                       // level L: Q = 0, X = Xin
                            if (SOS_PRECOMBINE_STREAM) { const double xm = q[KHM]; q[0] = xi + xm; q[KHM] = xi - xm; }
                            else *q = xi;
                        }
#pragma unroll 1
                        for (; cnt >= 8; cnt -= 8) fix_block<8, FS, NS, SOS_PRECOMBINE_STREAM ? KHM : 0>(q, qa, P, xi);
                        if (cnt & 4) fix_block<4, FS, NS, SOS_PRECOMBINE_STREAM ? KHM : 0>(q, qa, P, xi);
                        if (cnt & 2) fix_block<2, FS, NS, SOS_PRECOMBINE_STREAM ? KHM : 0>(q, qa, P, xi);
                        if (cnt & 1) fix_block<1, FS, NS, SOS_PRECOMBINE_STREAM ? KHM : 0>(q, qa, P, xi);
                    }
                    __syncthreads();
                    PH(1);
                    // source function of order ig for the levels of the chunk (SOS_FSOURCE_ORDREIG)
                    v4d acc[2][RTWH][CT];       // (zero for a bin without aerosol operator; the dense pass starts from the MFMA zero operand)
#pragma unroll
                    for (int sy = 0; sy < 2; sy++)
#pragma unroll
                        for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                            for (int ct = 0; ct < CT; ct++) acc[sy][rt][ct] = (v4d){0., 0., 0., 0.};
                    auto contract = [&](auto na_tag) {
                        constexpr int NA = decltype(na_tag)::value;
#define SOS_GEMM(RAYV, FOLDV)                                                                                          \
    gemm_source<NA, RAYV, FOLDV, RTWH, CT, NW, FS, KHM, true, (SOS_PRECOMBINE_STREAM != 0)>(acc, mpa, has_aer != 0, vtp, ufp, cx.ks2h, cx.rtph, bx,            \
                                                        cxdel + 1, cydel + 1, lane, wv, cx.prow, pcb)
                        if (s > 2) SOS_GEMM(-1, false);
                        else if (s & 1) { if (fold) SOS_GEMM(1, true); else SOS_GEMM(1, false); }
                        else { if (fold) SOS_GEMM(0, true); else SOS_GEMM(0, false); }
#undef SOS_GEMM
                    };
                    if (tile_b) contract(std::integral_constant<int, RTWH>());
                    else if (tile_a) contract(std::integral_constant<int, 1>());
                    else if (fold) __syncthreads();
                    __syncthreads();             // every wave has read the chunk
                    PH(2);
                    write_back_source<RTWH, CT, NW, FS, KHM>(acc, cbuf, lane, wv, KH, (s > 2 && has_aer) ? cxdel + 1 : nullptr);
                    __syncthreads();
                    PH(3);
                }
                // formal solution of the chunk, in place (SOS_INTEGR_EPOPT)
                if (active && !up) {
                    lds_f64 *q;
                    const lds_f64 *qa, *qd, *lx;
                    int cnt;
                    lds_f64 *const cb3 = (lds_f64 *)cbuf;
                    const lds_f64 *const ca3 = (const lds_f64 *)catt, *const ci3 = (const lds_f64 *)cidt, *const cx3 = (const lds_f64 *)ccxd;
                    if (chk == 0) {               // the ray enters at the top: X-(0) = 0, S-(0) is the first "previous" source
                        q = cb3 + rl; qa = ca3 + NS + jj; qd = ci3 + 1; lx = cx3 + 1;
                        dn_z = 0.;
                        dn_s = O1 ? o1.sva * lx[0] + o1.svr * lx[VL] : *q;       // no reflected-beam term at level 0 (SOS_OS.F:3280)
                        *q = 0.;
                        cnt = nlev - 1;
                    } else {
                        q = cb3 - FS + rl; qa = ca3 + jj; qd = ci3; lx = cx3; cnt = nlev;
                    }
#pragma unroll 1
                    for (; cnt >= 8; cnt -= 8) scan_block<1, 8, FS, NS, O1>(q, qa, qd, mu, dn_z, dn_s, lx, VL, o1);
                    if (cnt & 4) scan_block<1, 4, FS, NS, O1>(q, qa, qd, mu, dn_z, dn_s, lx, VL, o1);
                    if (cnt & 2) scan_block<1, 2, FS, NS, O1>(q, qa, qd, mu, dn_z, dn_s, lx, VL, o1);
                    if (cnt & 1) scan_block<1, 1, FS, NS, O1>(q, qa, qd, mu, dn_z, dn_s, lx, VL, o1);
                    if (L == nt) { xb = dn_z; gnd[SURF ? kk : c * NS + jj] = xb * usign; }
                    if (ZO && jout) {
                        if (jlo >= l0 && jlo <= L) xlo = cbuf[(size_t)(jlo - l0) * FS + rl];
                        if (jhi >= l0 && jhi <= L) xhi = cbuf[(size_t)(jhi - l0) * FS + rl];
                    }
                } else if (active) {
                    // up-going rows: the chunk with zero inflow at its bottom level L (Q_L = 0), then the link to level l0-1
                    lds_f64 *q = (lds_f64 *)(cbuf + (size_t)(L - l0) * FS + rl);
                    const lds_f64 *qa = (const lds_f64 *)(catt + (size_t)(L - l0) * NS + jj);          // layer L-1
                    const lds_f64 *qd = (const lds_f64 *)(cidt + (L - l0));
                    const lds_f64 *lx = (const lds_f64 *)(ccxd + (L - l0 + 1));                        // level L
                    double sn;
                    if (O1) {
                        sn = o1.sva * lx[0] + o1.svr * lx[VL];
                        if (o1.fres && L != nt) sn = sn + (o1.sfa * lx[2 * VL] + o1.sfr * lx[3 * VL]);   // SOS_OS.F:3280: none at the ground
                    } else sn = *q;
                    const double sbot = sn;
                    double z = 0.;
                    *q = 0.;
                    int cnt = nlev - 1;
#pragma unroll 1
                    for (; cnt >= 8; cnt -= 8) scan_block<-1, 8, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, VL, o1);
                    if (cnt & 4) scan_block<-1, 4, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, VL, o1);
                    if (cnt & 2) scan_block<-1, 2, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, VL, o1);
                    if (cnt & 1) scan_block<-1, 1, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, VL, o1);
                    if (chk == 0) q_top = z;
                    else {
                        // A(c): one more layer step (layer l0-1, source of level l0-1 kept from the previous chunk) on Q_l0
                        const double a = catt[jj];
                        const double w = a * (mu * cidt[0] + 1.0) - 1.0;
                        *lane_ptr(acf + chk * KHM, kk8) = w * (sn - up_sprev) + (z + a * (up_sprev - z));
                    }
                    up_sprev = sbot;
                    if (ZO && jout) {
                        if (jlo >= l0 && jlo <= L) qlo = cbuf[(size_t)(jlo - l0) * FS + rl];
                        if (jhi >= l0 && jhi <= L) qhi = cbuf[(size_t)(jhi - l0) * FS + rl];
                    }
                }
                __syncthreads();
                if (!O1) PH(4);
                // the chunk [Q+ | X-] of this order goes back to the scratch (levels l0..L only).  After the barrier above nobody
                // reads the attenuation / level-vector copies any more, and each 16-byte unit of the chunk is read here by ONE
                // thread -- the same one the LDS-DMA mapping gives it to: so the next chunk is requested unit by unit as soon as
                // this thread has sent the old content on its way, with no barrier in between, and its latency runs behind the
                // store pass instead of in front of the next contraction.
                {
                    const bool pre = SOS_STREAM_PREFETCH && !O1 && chk + 1 < nchunk;
                    const int units = nlev * FS / 2;
                    const v2d *src = reinterpret_cast<const v2d *>(cbuf);
                    v2d *dst = reinterpret_cast<v2d *>(fld + (size_t)l0 * FS);
                    const double *gnx = fld + (size_t)(l0 + COLS) * FS;
                    auto fetch = [&](int u) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gnx + 2 * (size_t)u),
                                                         (__attribute__((address_space(3))) void *)(cbuf + 2 * (size_t)(u - (t & 63))), 16, 0,
                                                         SOS_STREAM_NT);
                    };
                    if (pre) {
                        const int ln = l0 + COLS;
                        glds_copy<NTH, 0>(att + (size_t)ln * NS, catt, COLS * NS / 2, t);
                        stage_vec(ln);
                    }
                    // wave-uniform trip count; global address = scalar base + one of two per-lane 32-bit offsets (the 4 KiB
                    // step between consecutive units of a thread exceeds the immediate range, so odd units use v1 = v0 + 16 NTH
                    // and every second pair a second scalar base): no vector address arithmetic (lane_ptr, sos_dev.h)
                    const int full = units / NTH, rem = units - full * NTH;
                    char *db = reinterpret_cast<char *>(dst);
                    unsigned v0 = (unsigned)t * 16u, v1 = (unsigned)t * 16u + (unsigned)(NTH * 16);
                    const v2d *sp = src + t;
                    auto put = [&](const v2d &a, char *base, unsigned &vo) {
                        v2d *d = reinterpret_cast<v2d *>(lane_ptr(base, vo));
#if SOS_STREAM_NT
                        __builtin_nontemporal_store(a, d);
#else
                        *d = a;
#endif
                    };
                    int i = 0;
#pragma unroll 1
                    for (; i + 4 <= full; i += 4) {
                        const v2d a0 = sp[(i + 0) * NTH], a1 = sp[(i + 1) * NTH], a2 = sp[(i + 2) * NTH], a3 = sp[(i + 3) * NTH];
                        size_t o0 = (size_t)i * (NTH * 16), o1 = o0 + 2 * (NTH * 16);
                        asm volatile("" : "+s"(o1));
                        put(a0, db + o0, v0); put(a1, db + o0, v1); put(a2, db + o1, v0); put(a3, db + o1, v1);
                        if (pre) { fetch(t + i * NTH); fetch(t + (i + 1) * NTH); fetch(t + (i + 2) * NTH); fetch(t + (i + 3) * NTH); }
                    }
#pragma unroll 1
                    for (; i < full; i++) {
                        const v2d a0 = sp[i * NTH];
                        put(a0, db + (size_t)i * (NTH * 16), v0);
                        if (pre) fetch(t + i * NTH);
                    }
                    if (t < rem) {
                        const v2d a0 = sp[full * NTH];
                        put(a0, db + (size_t)full * (NTH * 16), v0);
                        if (pre) fetch(t + full * NTH);
                    }
                }
                if (O1) PH(7); else PH(5);
            }
            // after the last chunk: inflow of every chunk from the ground value upwards, Xin(c-1) = A(c) + B(c) Xin(c)
            if (active && up) {
                double x = bcv;
                int cq = nchunk - 1;
#pragma unroll 1
                while (cq >= 1) {
                    double av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int cc = max(cq - u, 1);
                        av[u] = *lane_ptr(acf + cc * KHM, kk8); bv[u] = *lane_ptr(bcf + cc * NS, jj8);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (cq - u >= 1) {
                            *lane_ptr(xin + (cq - u) * KHM, kk8) = x;
                            if (ZO && jout) { if (cq - u == jlo / COLS) xlo = qlo + plo * x; if (cq - u == jhi / COLS) xhi = qhi + phi * x; }
                            x = av[u] + bv[u] * x;
                        }
                    cq -= 8;
                }
                *lane_ptr(xin, kk8) = x;
                if (ZO && jout) { if (jlo / COLS == 0) xlo = qlo + plo * x; if (jhi / COLS == 0) xhi = qhi + phi * x; }
                xb = q_top + *lane_ptr(bcf, jj8) * x;
            }
        };

        // ---- scattering orders ------------------------------------------------------------------------------------------
        double i3 = 0., a1 = 0., d1 = 0., g1 = 0.;
        double i3lo = 0., dlo = 0., i3hi = 0., dhi = 0.;
        int ig = 1, iglast = 1;
        for (;;) {
            if (ig == 1) pass(std::true_type(), bc);                                 // SOS_OS.F:1025
            else pass(std::false_type(), bc);                                        // SOS_OS.F:1157,1244
            if (ig == 1) {                                                           // SOS_OS.F:1094-1137
                __syncthreads();
                i3 = xb; a1 = 0.; d1 = xb; g1 = 0.;
                if (ZO) { i3lo = xlo; dlo = xlo; i3hi = xhi; dhi = xhi; }
                bc = ground_bc();
            } else {
                g1 = xb;
                const double i3n = i3 + g1;
                const double ag = fabs(g1);
                const bool p1 = active && ig != 2 && conv_exceeds(a1, d1, g1, i3, cx.thr_cv);       // SOS_PARAM_CONV
                const bool p2 = active && ag > cx.thr_val;                                          // SOS_ARRET_DIFFUS_1
                const bool p4 = active && i3n != 0.0 && ag > cx.thr_sum * fabs(i3n);                // SOS_ARRET_DIFFUS_2
                const int pm = block_or_bits<NW>(p1, p2, p4, reinterpret_cast<int *>(red), wv, lane, red_slot);
                bc = ground_bc();
                if (ig != 2 && !(pm & 1)) {                                          // SOS_OS.F:1293-1315
                    i3 = i3 + queue_term(d1, g1);
                    if (ZO) { i3lo = i3lo + queue_term(dlo, xlo); i3hi = i3hi + queue_term(dhi, xhi); }
                    break;
                }
                a1 = d1; d1 = g1;                                                    // SOS_OS.F:1323-1363
                i3 = i3n;
                if (ZO) { dlo = xlo; dhi = xhi; i3lo = i3lo + xlo; i3hi = i3hi + xhi; }
                if (!(pm & 2)) break;                                                // SOS_OS.F:1370
                if (!(pm & 4)) break;                                                // SOS_OS.F:1389
                if (!(ig < cx.igmax)) break;                                         // SOS_OS.F:1406
            }
            ig = ig + 1;
            if (ig > cx.igmax) break;
            iglast = ig;
        }
        // SOS_OS.F:1421-1439 (see sos_os.hip: record from I3OUT - RIIOUT, tests and fluxes from I3 - RII)
        double i3out0 = i3;
        if (SURF && active && up) {                                                  // SOS_OS.F:1062-1084
            const double rii = exp(-htot / mu) * dirterm;
            const double riilo = exp(-(htot - ((ZO && jout) ? hlo : h0)) / mu) * dirterm;
            i3out0 = i3 - riilo; i3 = i3 - rii;
            if (ZO) {
                const double riihi = (jout ? exp(-(htot - hhi) / mu) : 0.) * dirterm;
                i3lo = i3lo - riilo; i3hi = i3hi - riihi;
            }
        }
        if (s == 0) {                                                                // SOS_OS.F:1447-1456
            if (active && c == 0) i3s[up ? jj : NS + jj] = i3;
            __syncthreads();
            if (t == 0) {
                double em = 0., ep = 0.;
                for (int j = 0; j < N; j++) {
                    em = em + cx.mu[j] * cx.ga[j] * i3s[NS + j];
                    ep = ep + cx.mu[j] * cx.ga[j] * i3s[j];
                }
                bn.flux[2 * b] = em * 2 / cx.mus;
                bn.flux[2 * b + 1] = ep * 2 / cx.mus;
            }
        }
        const double coef = (s == 0) ? 1. : 2.;                                      // SOS_OS.F:1460-1473
        i4 = i4 + coef * i3;
        i5 = i5 + coef * i3 * sign;
        if (active) {                                                                // SOS_OS.F:1484-1534,1572
            const double outv = (ZO && jout) ? ((1 - zz) * i3lo + zz * i3hi) : i3out0;
            recb[(size_t)s * 3 * W + recoff] = outv * usign;
            if (up && jj == 0) recb[(size_t)s * 3 * W + c * W + N] = 0.;
        }
        if (t == 0) bn.iglast[(size_t)b * S1 + s] = iglast;
        nord = s + 1;
        // (The order loop is left by its regular exits also in this form: the test below then runs on partial sums and its
        //  result is not used.  A `break` right here (-DSOS_EXP_SPEC_BREAK) makes the <4,2,ZO,SURF> instantiation -- 365 spilled
        //  SGPRs, 86 spilled VGPRs in that build -- store I3 = 0 for every DOWN-going row (waves 2, 3) and a wrong value for
        //  thread 0, so that the replay stops the series after 9 orders instead of 31: reproduced and dumped with
        //  scripts/spec_break_probe.py (round 3).  The dump is deterministic and wave-uniform (all of waves 2 and 3, nothing of
        //  waves 0 and 1 but thread 0), so it is not a race; the source has no path on which a down-going row's I3 is zero
        //  there; in that build's ISA the store, its address (SGPR pair restored from VGPR lanes) and its exec mask are right
        //  and the stored register pair is one the allocator shares between xb and i3 -- where its last definition for the
        //  down-going waves is lost was not traced through the 24 000 lines (a build without SGPR spills into VGPR lanes,
        //  -mllvm -amdgpu-spill-sgpr-to-vgpr=0, would isolate that mechanism: the backend refuses it for this kernel,
        //  "unhandled SGPR spill to memory").  The shipped exit keeps I3 live into the stop
        //  test, which every instantiation needs anyway, and the hand-over is checked against the per-bin launch for ALL
        //  twelve (NW, RTWH, ZO, SURF) instantiations with several rounds: tests/test_gpu_parity.py, _ALL_VARIANTS.)
        if (spec) bn.spec_i3[((size_t)b * S1 + s) * NTH + t] = i3;
#ifdef SOS_EXP_SPEC_BREAK
        if (spec) break;          // diagnostic build: the exit that made <4,2,ZO,SURF> hand over wrong I3 terms
#endif
        PH(6);
        const double a3 = fabs(i3);                                                                 // SOS_ARRET_FOURIER
        const bool pf = active && ((i4 != 0.0 && a3 > cx.thr_sf * fabs(i4)) || (i5 != 0.0 && a3 > cx.thr_sf * fabs(i5)));
        const int pf2 = block_or_bits<NW>(pf, reinterpret_cast<int *>(red), wv, lane, red_slot);
        if (!pf2 || s == iborm) { finished = true; break; }                          // SOS_OS.F:1585
    }
    if (spec) return false;
    if (finished) {
        for (int i = t + nord; i < S1; i += NTH) bn.iglast[(size_t)b * S1 + i] = 0;
        if (t == 0) { bn.norders[b] = nord; state[0] = 1.; }
    } else {                                             // more orders follow in the next launch
        state[8 + 2 * t] = i4; state[8 + 2 * t + 1] = i5;
        if (t == 0) bn.norders[b] = nord;
    }
#ifdef SOS_PROFILE_PHASES
    // SOS_PHASE_ROLE = 1 / 2: stamps of the up-going / down-going waves only (diagnostic: skew between the two halves)
#ifndef SOS_PHASE_ROLE
#define SOS_PHASE_ROLE 0
#endif
    if (bn.phase && lane == 0 && (SOS_PHASE_ROLE == 0 || (SOS_PHASE_ROLE == 1) == up))
        for (int k = 0; k < 8; k++) atomicAdd(&bn.phase[(size_t)b * 8 + k], ph_acc[k]);
#endif
    return finished;
    };   // run_task

    if (!PERSIST) {
        // Order-parallel form for FEW bins (a band of 1-100 bins leaves most of the chip idle while every bin walks through
        // its Fourier orders one after the other -- 11 ms for a single bin): the orders of a bin do not depend on each
        // other, only the Fourier stop test does (running sums I4, I5 over the orders, SOS_OS.F:1460-1473,1585).  bn.spec_k =
        // -K: set-up launch, one workgroup per bin writes the bin's constants into region b K.  bn.spec_k = K > 0: workgroup
        // (b, j) runs order s_begin + j of bin b in its own work region b K + j and stores its I3 terms; k_sos_stream_replay
        // then replays the stop tests of these K orders in sequence.  Orders past the stop are wasted work on idle CUs.
        const int K = bn.spec_k;
        if (K < 0) run_task(blockIdx.x, 0, 0, (size_t)blockIdx.x * (size_t)(-K), (size_t)blockIdx.x * (size_t)(-K), false);
        else if (K > 0) {
            const int b = blockIdx.x / K, j = blockIdx.x - b * K;
            if (bn.s_begin + j < bn.s_end) run_task(b, bn.s_begin + j, bn.s_begin + j + 1, (size_t)b * K, (size_t)b * K + j, true);
        } else { const int b = SOS_BIN_INDEX(bn); run_task(b, bn.s_begin, bn.s_end, b, b, false); }
        return;
    }
    // ---- persistent form: order-scheduled tasks ---------------------------------------------------------------------------
    // Every workgroup streams the 240 KB source operator of its Fourier order once per chunk; with one workgroup per bin the
    // 64 workgroups of an XCD sit at ~20 different orders and a fifth of those reads miss the 4 MB L2 (a quarter of the
    // kernel's fabric traffic).  Here the bins are dealt to the XCDs (bin b -> queue b mod 8) and the workgroups of an XCD --
    // which one it is read from HW_REG_XCC_ID -- take tasks (order s, bin) from their XCD's queue in the order s-major: at any
    // time they work on one or two orders and share their operators in L2.  The queue is one atomic counter; task c of a queue
    // with nq bins is (s = c / nq, bin = c mod nq).  A task starts when its bin has completed order s - 1: flag[bin] counts the
    // completed orders (FIN: the series has ended), published with an agent-scope release by the workgroup that ran them and
    // acquired by the taker -- a workgroup only ever waits for a task taken earlier from the same queue by a workgroup that
    // is running, so the waits cannot cycle.  A workgroup whose queue is exhausted takes tasks from the other queues (also what
    // makes the result independent of where the hardware places workgroups).
    constexpr int FIN = 0x7fffffff;
    const int t = threadIdx.x, S1 = cx.smax + 1;
    int *ired = reinterpret_cast<int *>(red) + 24;             // task broadcast (block_or_* use the first 2 NW ints of red)
    const int my_xcd = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7;   // hwreg(HW_REG_XCC_ID, 0, 4)
    const int nbins = bn.nb;
    for (int hop = 0; hop < 8; hop++) {
        const int q = (my_xcd + hop) & 7;
        const int nq = (nbins - q + 7) >> 3;                   // bins q, q + 8, ... of this queue
        if (nq <= 0) continue;
        for (;;) {
            __syncthreads();                                   // the previous task's last LDS reads / ired reads are done
            if (t == 0) {
                int tb = -1, ts = 0;
                for (;;) {
                    // all bins of the queue finished -> nothing left here (bins in flight are finished by their holders)
                    if (__hip_atomic_load(&bn.queue[16 * (8 + q)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nq) break;
                    const int c = __hip_atomic_fetch_add(&bn.queue[16 * q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ts = c / nq;
                    if (ts >= S1) break;
                    const int cand = q + 8 * (c - ts * nq);
                    int f;
                    while ((f = __hip_atomic_load(&bn.qflag[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < ts)
                        __builtin_amdgcn_s_sleep(8);
                    if (f != FIN) { tb = cand; break; }
                }
                int te = ts + 1;
                if (tb >= 0 && nq - __hip_atomic_load(&bn.queue[16 * (8 + q)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= bn.q_tail) {
                    // the tail of the queue: fewer unfinished bins than workgroups to run them order by order -- this
                    // workgroup keeps the bin for all its remaining orders; nobody else will touch it (flag = FIN at once)
                    te = S1;
                    __hip_atomic_store(&bn.qflag[tb], FIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&bn.queue[16 * (8 + q)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ired[0] = tb; ired[1] = ts; ired[2] = te;
            }
            __syncthreads();
            const int tb = uniform_i32(ired[0]), ts = uniform_i32(ired[1]), te = uniform_i32(ired[2]);
            if (tb < 0) break;
            const bool fin = run_task(tb, ts, te, tb, tb, false);
            if (te == S1) continue;                            // kept to the end: nothing to hand over
            // publish: every wave's stores have left the CU, then one lane releases and raises the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&bn.qflag[tb], fin ? FIN : ts + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fin) __hip_atomic_fetch_add(&bn.queue[16 * (8 + q)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Order-parallel form, second half: the Fourier stop test (SOS_ARRET_FOURIER, SOS_OS.F:3709; accumulation :1460-1473, exit
// :1585) of orders [s0, s1) of every bin, replayed in sequence from the I3 terms the order tasks left in bn.spec_i3 -- the same
// statements as at the end of run_task's order loop, with the same thread -> row mapping.  One workgroup per bin.
template <int NW, int KHT>
__global__ __launch_bounds__(64 * NW) void k_sos_stream_replay(const SosDev cx, const SosBins bn, int s0, int s1)
{
    __shared__ int red[2 * NW];
    constexpr int NTH = 64 * NW, HW = NW / 2, KHM = skhm(KHT), FS = sfs(KHT), NS = sns(KHT);
    const int LPB = bn.lpb, NCH = LPB / COLS, VS = LPB + VPAD, N = cx.n, S1 = cx.smax + 1;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    if (uniform_i32(bn.norders[b]) < 0) return;                     // malformed bin (flagged by the set-up launch)
    double *state = bn.scratch + (size_t)b * (size_t)bn.spec_k * bn.scr_stride + (size_t)LPB * FS + (size_t)(LPB + 1) * NS +
                    (size_t)7 * VS + (size_t)NCH * (2 * KHM + 2 * NS);
    if (uniform_f64(state[0]) != 0.) return;                        // finished in an earlier round
    const int iborm = uniform_i32(bn.iborm[b]);
    const bool up = wv < HW;
    const bool active = (up ? t : t - 64 * HW) < 3 * N;
    double i4 = state[8 + 2 * t], i5 = state[8 + 2 * t + 1];
    int nord = s0, red_slot = 0;
    bool finished = s0 > iborm;
    for (int s = s0; s < s1 && !finished; ++s) {
        const double sign = (s & 1) ? -1. : 1.;
        const double i3 = bn.spec_i3[((size_t)b * S1 + s) * NTH + t];
        const double coef = (s == 0) ? 1. : 2.;
        i4 = i4 + coef * i3;
        i5 = i5 + coef * i3 * sign;
        nord = s + 1;
        const double a3 = fabs(i3);
        const bool pf = active && ((i4 != 0.0 && a3 > cx.thr_sf * fabs(i4)) || (i5 != 0.0 && a3 > cx.thr_sf * fabs(i5)));
        const int pf2 = block_or_bits<NW>(pf, red, wv, lane, red_slot);
        if (!pf2 || s == iborm) finished = true;
    }
    if (finished) {
        for (int i = t + nord; i < S1; i += NTH) bn.iglast[(size_t)b * S1 + i] = 0;
        // the order tasks of this round past the stop have stored their records: sosgpu.h promises that only orders
        // 0 .. norders-1 are written (a zero-filled d_rec stays zero beyond them), so those rows are cleared again
        const int W3 = 3 * cx.w, s_hi = min(s1, iborm + 1);
        double *recb = bn.rec + (size_t)b * S1 * W3;
        for (int i = nord * W3 + t; i < s_hi * W3; i += NTH) recb[i] = 0.;
        if (t == 0) { bn.norders[b] = nord; state[0] = 1.; }
    } else {
        state[8 + 2 * t] = i4; state[8 + 2 * t + 1] = i5;
        if (t == 0) bn.norders[b] = nord;
    }
}

// ---------------------------------------------------------------------------------------------
static size_t stream_lds_bytes(int kht)
{
    const int fs = sfs(kht), ns = sns(kht);
    return ((size_t)COLS * fs + 3 * ns + 2 + 2 * ns + 16 + 2 * ns + (size_t)COLS * ns + 7 * (COLS + VPAD)) * sizeof(double);
}

// waves, row tiles per wave, row tiles of the layout for N directions
static void stream_shape(int n, int *nw, int *rtw, int *kht)
{
    const int kh = sos_round_up(3 * n, 8);
    if (kh > 128) { *nw = 8; *rtw = 2; *kht = 16; }
    else if (kh <= 64) { *nw = 4; *rtw = 1; *kht = 4; }
    else {
        *nw = 4; *rtw = 2;
        *kht = kh <= 80 ? 5 : kh <= 96 ? 6 : 8;
#ifdef SOS_STREAM_WIDE
        *kht = 8;                                          // round 2's layout (A/B measurements)
#endif
        // the persistent, order-scheduled form (opt-in) exists for the full-width layout only
        if (const char *e = getenv("SOSGPU_STREAM_PERSIST")) { if (atoi(e) != 0) *kht = 8; }
    }
}

#ifndef SOS_MULTI
size_t sos_stream_scratch_doubles(int n, int lpb)
{
    int nw, rtw, kht;
    stream_shape(n, &nw, &rtw, &kht);
    return stream_scratch_doubles(nw, kht, lpb);
}
#endif

template <int NW, int RTWH, bool ZO, bool SURF, int KHT = NW * RTWH>
static int launch_stream_variant(const SosDev &cx, const SosBins &bn, hipStream_t st, int *hip_err)
{
#ifdef SOS_MULTI
    const bool persist = false;                            // (the per-bin context is bound to blockIdx.x)
    if (bn.queue) return SOSGPU_E_UNSUPPORTED;
    auto kern = k_sos_stream<NW, RTWH, ZO, SURF, false, KHT>;
#else
    const bool persist = bn.queue != nullptr;
    if (persist && KHT != NW * RTWH) return SOSGPU_E_UNSUPPORTED;       // (stream_shape keeps the full width for that form)
    auto kern = k_sos_stream<NW, RTWH, ZO, SURF, false, KHT>;
    if constexpr (KHT == NW * RTWH) { if (persist) kern = k_sos_stream<NW, RTWH, ZO, SURF, true, KHT>; }
#endif
    const size_t lds = stream_lds_bytes(KHT);
    // the dynamic-LDS limit of a kernel is set once per device and size (the call costs tens of microseconds: with few bins per
    // wavelength the host launch path is what bounds a hyperspectral loop, scripts/spectrum_bench.py)
    // (atomics: host threads of run_sos.sos_proc_many launch concurrently; a lost update only repeats the call)
    static std::atomic<size_t> configured[2][16];           // (per instantiation of this function: one per kernel variant)
    static std::atomic<int> cus[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && (dev < 0 || dev >= 16 || configured[persist][dev].load(std::memory_order_acquire) < lds)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && dev >= 0 && dev < 16) configured[persist][dev].store(lds, std::memory_order_release);
    }
    int grid = bn.spec_k > 0 ? bn.nb * bn.spec_k : bn.nb;
    if (e == hipSuccess && persist) {
        // as many workgroups as the chip hosts at once (two per CU for the 4-wave forms); fewer resident ones only cost speed
        int ncu = (dev >= 0 && dev < 16) ? cus[dev].load(std::memory_order_relaxed) : 0;
        if (!ncu) {
            e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess && dev >= 0 && dev < 16) cus[dev].store(ncu, std::memory_order_relaxed);
        }
        grid = std::min(bn.nb, ncu * (NW == 4 ? 2 : 1));
    }
    SosBins bq = bn;
    if (persist && bq.q_tail < 0) bq.q_tail = 2 * ((grid + 7) / 8);     // unfinished bins per queue below which bins are kept
    if (e == hipSuccess) {
        kern<<<grid, 64 * NW, lds, st>>>(cx, bq);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { if (hip_err) *hip_err = (int)e; return -2; }
    return 0;
}

#ifndef SOS_MULTI
int sos_stream_threads(int n)
{
    int nw, rtw, kht;
    stream_shape(n, &nw, &rtw, &kht);
    return 64 * nw;
}

int launch_sos_stream_replay(const SosDev &cx, const SosBins &bn, int s0, int s1, hipStream_t st, int *hip_err)
{
    int nw, rtw, kht;
    stream_shape(cx.n, &nw, &rtw, &kht);
    if (kht == 4) k_sos_stream_replay<4, 4><<<bn.nb, 256, 0, st>>>(cx, bn, s0, s1);
    else if (kht == 5) k_sos_stream_replay<4, 5><<<bn.nb, 256, 0, st>>>(cx, bn, s0, s1);
    else if (kht == 6) k_sos_stream_replay<4, 6><<<bn.nb, 256, 0, st>>>(cx, bn, s0, s1);
    else if (kht == 8) k_sos_stream_replay<4, 8><<<bn.nb, 256, 0, st>>>(cx, bn, s0, s1);
    else k_sos_stream_replay<8, 16><<<bn.nb, 512, 0, st>>>(cx, bn, s0, s1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { if (hip_err) *hip_err = (int)e; return -2; }
    return 0;
}
#endif

int launch_sos_stream(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err)
{
    if (cx.n < 1 || cx.n > 85 || nt_max < 1 || nt_max > 1023) return SOSGPU_E_UNSUPPORTED;
    int nw, rtw, kht;
    stream_shape(cx.n, &nw, &rtw, &kht);
    if (cx.kh > skhm(kht) || cx.rtph * 16 < cx.kh || cx.rtph > kht) return SOSGPU_E_UNSUPPORTED;
    if (!bn.scratch || bn.lpb < nt_max + 1 || bn.lpb % COLS || bn.scr_stride < stream_scratch_doubles(nw, kht, bn.lpb))
        return SOSGPU_E_UNSUPPORTED;
    const int zo = bn.jout != nullptr;
#define V(NWV, R, K)                                                                           \
    if (nw == NWV && rtw == R && kht == K) {                                                   \
        if (cx.imat_surf)                                                                      \
            return zo ? launch_stream_variant<NWV, R, true, true, K>(cx, bn, st, hip_err)      \
                      : launch_stream_variant<NWV, R, false, true, K>(cx, bn, st, hip_err);    \
        return zo ? launch_stream_variant<NWV, R, true, false, K>(cx, bn, st, hip_err)         \
                  : launch_stream_variant<NWV, R, false, false, K>(cx, bn, st, hip_err);       \
    }
    V(4, 1, 4) V(4, 2, 5) V(4, 2, 6) V(4, 2, 8) V(8, 2, 16)
#undef V
    return SOSGPU_E_UNSUPPORTED;
}
