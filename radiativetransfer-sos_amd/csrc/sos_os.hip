// csrc/sos_os.hip -- fused successive-orders-of-scattering solver for a batch of CKD bins (gfx950).
//
// Replaces, for every bin of a wavelength at once, the reference call chain
//   SOS_PROC.F:3459-3594 (bin loop) -> SOS.F:554 -> SOS_OS (src/SOS_OS.F:303-1674)
// with its leaves SOS_FSOURCE_ORDRE1 (:2431), SOS_FSOURCE_ORDREIG (:2663), SOS_INTEGR_EPOPT (:2222),
// SOS_FSOURCE_DIFF_FRESNEL1 (:3106), the stop tests (:3377,:3497,:3585,:3709) and SOS_AJOUT_QUEUE (:3871).
//
// MI355X mapping (one 256-thread workgroup = one bin, all Fourier orders s and scattering orders ig):
//  * the radiance field L(6N rows x NT+1 levels) of the current scattering order lives in LDS for the
//    whole solve, laid out [level][row] (row contiguous) with a level stride of KP+2 doubles;
//  * the order-ig source S = M_s (XDEL o L) [+ M_ray,s (YDEL o L) for s <= 2] is a dense
//    [6N x 6N] x [6N x (NT+1)] FP64 contraction run on v_mfma_f64_16x16x4_f64: each of the 4 waves owns
//    RTW row tiles x CT column tiles of accumulators (registers); the operator M_s streams from L2/MALL
//    in pre-packed A-fragment order (1 KiB contiguous per wave load), the B fragments come from LDS and
//    are scaled by XDEL/YDEL of their level in registers;
//  * S overwrites L in LDS, then one thread per (Stokes, direction) row runs the layer-by-layer formal
//    solution (SOS_INTEGR_EPOPT) in place, using attenuations exp(-dtau/mu) precomputed once per bin;
//  * stop tests are three max-reductions per scattering order fused into one wavefront-shuffle +
//    LDS reduction; the geometric-series tail, the Fourier accumulation and the Fourier stop are
//    per-thread register state.
// HBM traffic per bin: 3(NT+1) doubles in, (F*3*(2N+1) + small) doubles out; everything else is on chip
// or L2/MALL-resident operator reads shared by all bins.
#include <cstdlib>
#include "sos_common.h"
#include "kernels.h"

#define SOSGPU_E_UNSUPPORTED -3
#ifndef SOS_SCAN_UNROLL
#define SOS_SCAN_UNROLL 4
#endif
#ifdef SOS_PROFILE_PHASES
#define PH_T0() unsigned long long ph_t = __builtin_amdgcn_s_memtime()
#define PH(k) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define PH_T0() do {} while (0)
#define PH(k) do {} while (0)
#endif
#define SOS_PRAGMA(x) _Pragma(#x)
#define SOS_UNROLL(n) SOS_PRAGMA(unroll n)

// Force a value the whole wave agrees on into scalar registers, so that the loop exits it decides are
// uniform branches (keeps s / ig / operator pointers in SGPRs instead of per-lane VGPRs).
__device__ __forceinline__ double uniform_f64(double v)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// bitwise OR over the workgroup of up to 3 predicate bits (red: 4 ints of LDS); two barriers, no FP64 work.
// (__syncthreads_or only returns a logical OR.)
__device__ __forceinline__ int block_or_bits(int bits, int *red)
{
    int w = 0;
    if (__ballot(bits & 1)) w |= 1;
    if (__ballot(bits & 2)) w |= 2;
    if (__ballot(bits & 4)) w |= 4;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    const int r = red[0] | red[1] | red[2] | red[3];
    __syncthreads();
    return uniform_i32(r);
}

// FP64 vector instructions share the FP64 datapath with v_mfma_f64 on gfx950 (equal peak rates): every f64 VALU
// instruction issued while the co-resident workgroup is contracting waits for a 64-cycle MFMA slot.  The stop
// tests are therefore evaluated WITHOUT divisions and WITHOUT floating-point reductions: "max_k |y_k| > thr" is
// the same decision as "exists k: |num_k| > thr |den_k|", reduced with one __syncthreads_or of predicate bits.
//
// SOS_PARAM_CONV (SOS_OS.F:3434-3453): y = ((g/d - d/a) / (1 - g/d)^2) (g/x3) = (g a - d^2) d g / (a (d-g)^2 x3)
// for a, d, x3 != 0.  (Differs from the quotient form only by rounding at the 1e-16 level of a 1e-5 threshold;
// operands below ~1e-77 underflow in the products and are then ignored -- they are 1e-60 of the radiance scale.)
__device__ __forceinline__ bool conv_exceeds(double a, double d, double g, double x3, double thr)
{
    if (a != 0.0 && d != 0.0 && x3 != 0.0) {
        const double dg = d - g;
        const double num = (g * a - d * d) * (d * g);
        const double den = (a * x3) * (dg * dg);
        return fabs(num) > thr * fabs(den);
    }
    return false;
}
// SOS_AJOUT_QUEUE (SOS_OS.F:3959-3975): g / (1 - g/d) = g d / (d - g)
__device__ __forceinline__ double queue_term(double d, double g)
{
    return (d == 0.) ? 0. : (g * d) / (d - g);
}

// Instruction budget.  Two workgroups share a CU, one wave of each per SIMD.  While one wave streams
// v_mfma_f64 (64 cycles each, back to back) its partner gets only the left-over vector-issue slots, and every
// vector instruction of either wave delays the next MFMA.  Measured (phase stamps, profiles/): every non-MFMA VALU
// instruction costs ~12-15 cycles in this regime whatever its kind, so the non-contraction phases are written for
// the LOWEST VALU INSTRUCTION COUNT (running pointers, immediate offsets, no predicates, no divisions), not for
// latency.
//
// Field storage convention: rows [0,KH) hold X(+mu), rows [KH,2KH) hold X(-mu) in half-system order
// kk = c*N + (k-1), and the U component of the down-going half is stored NEGATED (V- = -U(-mu)).  With that the
// parity combinations need no per-row sign:  X^A = X+ + X-,  X^B = X+ - X-,  S+ = E^A + E^B,  S- = E^A - E^B
// (for U: X^A_U = U+ - U-, stored S-_U = -(E^B - E^A)); the formal solution is linear with a zero boundary for
// down-going rows, so it maps a negated source to a negated field.  The sign is restored where U(-mu) leaves the
// field: ground values (gnd) and output records.
//
// Source contraction in the parity-decomposed form (sos_common.h): for both half systems
//   acc[sys] = XDEL o (M^sys X^sys)                                              (aerosol operator, dense)
// plus, for s <= 2, the molecular operator in its exact rank-4 form on the one half system it acts on:
//   acc[sr] += U (YDEL o (V^T X^sr))                                             (noyaux.hip k_pack_ray)
// Each wave owns RTWH row tiles of BOTH systems x CT column tiles.
template <int CT>
__device__ __forceinline__ void b_fragments(v2d (&ba)[CT], v2d (&bb)[CT], const double *const (&bp)[CT], int KH, int m)
{
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const v2d xp = *reinterpret_cast<const v2d *>(bp[ct] + 8 * m);
        const v2d xm = *reinterpret_cast<const v2d *>(bp[ct] + KH + 8 * m);
        ba[ct] = xp + xm;
        bb[ct] = xp - xm;
    }
}

template <int RTWH, int CT>
__device__ __forceinline__ void gemm_source(v4d (&acc)[2][RTWH][CT], const double *__restrict__ mp, bool do_aer,
                                            const double *__restrict__ vt, const double *__restrict__ uf, int ray_sys,
                                            int ks2h, int rtph, int n2, const double *fld, int CS, int KH,
                                            const double *xdel, const double *ydel, int lane, int wv)
{
    const size_t rts = (size_t)ks2h * 64;               // v2d stride between row tiles
    const size_t sys_stride = (size_t)rtph * rts;       // v2d stride between the two systems
    const double *bp[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) bp[ct] = fld + (size_t)(ct * 16 + (lane & 15)) * CS + 2 * (lane >> 4);
    if (do_aer) {
        const v2d *ap = reinterpret_cast<const v2d *>(mp) + ((size_t)(wv * RTWH) * ks2h) * 64 + lane;
        // two register sets of A fragments used alternately (k-pair m from one set while m+1 loads into the other):
        // no register-to-register copies in the loop
        v2d a0[2][RTWH], a1[2][RTWH];
        auto load_a = [&](v2d (&a)[2][RTWH], int m) {
#pragma unroll
            for (int sy = 0; sy < 2; sy++)
#pragma unroll
                for (int rt = 0; rt < RTWH; rt++) a[sy][rt] = ap[sy * sys_stride + rt * rts + (size_t)m * 64];
        };
        auto mma = [&](const v2d (&a)[2][RTWH], int m) {
            v2d ba[CT], bb[CT];
            b_fragments<CT>(ba, bb, bp, KH, m);
#pragma unroll
            for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    acc[0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][rt].x, ba[ct].x, acc[0][rt][ct], 0, 0, 0);
                    acc[1][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][rt].x, bb[ct].x, acc[1][rt][ct], 0, 0, 0);
                }
#pragma unroll
            for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    acc[0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][rt].y, ba[ct].y, acc[0][rt][ct], 0, 0, 0);
                    acc[1][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][rt].y, bb[ct].y, acc[1][rt][ct], 0, 0, 0);
                }
        };
        load_a(a0, 0);
        int m = 0;
#pragma unroll 1
        for (; m + 1 < ks2h; m += 2) {
            load_a(a1, m + 1);
            mma(a0, m);
            if (m + 2 < ks2h) load_a(a0, m + 2);
            mma(a1, m + 1);
        }
        if (m < ks2h) mma(a0, m);
        // XDEL of the output level: every accumulator register of a lane belongs to one column
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const double sc = xdel[ct * 16 + (lane & 15)];
#pragma unroll
            for (int sy = 0; sy < 2; sy++)
#pragma unroll
                for (int rt = 0; rt < RTWH; rt++) acc[sy][rt][ct] *= sc;
        }
    }
    if (ray_sys >= 0) {
        // projections pr = V^T X^sr: one 16-row tile (rows 0..3 used) per column tile, every wave computes them itself
        // (64 MFMAs instead of the 256 of a dense pass, and no cross-wave exchange)
        const v2d *vp = reinterpret_cast<const v2d *>(vt) + lane;
        v4d pr[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) pr[ct] = (v4d){0., 0., 0., 0.};
#pragma unroll 1
        for (int m = 0; m < ks2h; m++) {
            const v2d a = vp[(size_t)m * 64];
            v2d ba[CT], bb[CT];
            b_fragments<CT>(ba, bb, bp, KH, m);
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const v2d b = ray_sys ? bb[ct] : ba[ct];
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, pr[ct], 0, 0, 0);
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, pr[ct], 0, 0, 0);
            }
        }
        // register 0 of the result holds row (lane>>4) in 0..3, column lane&15: exactly the B-operand layout of one
        // K = 4 step, so the expansion U * (YDEL o pr) needs no data movement
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const double q = pr[ct][0] * ydel[ct * 16 + (lane & 15)];
#pragma unroll
            for (int rt = 0; rt < RTWH; rt++) {
                const double u = uf[(size_t)(wv * RTWH + rt) * 64 + lane];
                if (ray_sys) acc[1][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, q, acc[1][rt][ct], 0, 0, 0);
                else acc[0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, q, acc[0][rt][ct], 0, 0, 0);
            }
        }
    }
}

// BIG = false: the whole field (NT+1 <= 16*CT levels) lives in LDS.
// BIG = true : the field, the attenuation table and the level vectors live in a per-bin HBM/L2 scratch
//              (reference profiles have NT = 100..600, SOS.h:202,229); the contraction runs over chunks of
//              16*CT levels staged through LDS, the formal solution streams the scratch with batched loads.
// ZO = true : output at an intermediate altitude (ZOUT != -1, SOS_OS.F:1511-1534) -- tracks two extra levels per row.
// Register bound: 256 architectural VGPRs for every variant (two workgroups per CU when the LDS allows it).  The
// 512-register form (accumulators and spills in AGPRs) of the CT = 4 variants was measured slower than the bounded
// form with a few scratch spills (60.2k vs 63.6k bins/s at N = 41, NT = 60) and one of its instantiations
// (<2,4,false,true> with a 4-level scan block) produced wrong down-going rows on gfx950, so it is not used.
template <int RTWH, int CT, bool BIG, bool ZO>
__global__ __launch_bounds__(256, 2) void k_sos_os(const SosDev cx, const SosBins bn)
{
    extern __shared__ double smem[];
    constexpr int COLS = 16 * CT;
#ifdef SOS_SCAN_UNROLL_FORCE
    constexpr int SU = SOS_SCAN_UNROLL_FORCE;
#else
    constexpr int SU = 4;                        // formal-solution unroll
#endif
    const int N = cx.n, R6 = cx.r6, KP = cx.kp, KH = cx.kh, CS = 2 * cx.kh + 2, W = cx.w;
    const int LPB = BIG ? bn.lpb : COLS;   // level capacity of the field storage
    const int FS = BIG ? 2 * cx.kh : CS;   // level stride of the field storage
    double *cbuf = smem;                   // [COLS][CS]  LDS: the field itself, or the staging chunk (BIG)
    double *gnd = cbuf + COLS * CS;        // [3][N] down-going field at the ground, order ig-1
    double *i3s = gnd + 3 * N;             // [2N]   I3 of the I rows (flux integrals)
    double *red = i3s + 2 * N;             // [16]
    double *lga = red + 16;                // [N] Gauss weights, [N] mu (LDS copies for the ground-reflection sums)
    double *lmu = lga + N;
    double *sbase = BIG ? bn.scratch + (size_t)blockIdx.x * bn.scr_stride : lmu + N;
    double *fld = BIG ? sbase : cbuf;      // [LPB][FS]   field / source, [level][+mu rows | -mu rows]
    double *att = BIG ? sbase + (size_t)LPB * FS : sbase;   // [LPB][N] exp(-dtau_i/mu_j), layer i = levels i..i+1
    double *dtau = att + (size_t)LPB * N;  // [LPB] each:
    double *idtau = dtau + LPB;
    double *xdel = idtau + LPB;
    double *ydel = xdel + LPB;
    double *ch = ydel + LPB;
    double *fco = ch + LPB;
    double *hh = fco + LPB;

#ifdef SOS_STATIC_PRIO
    // Two workgroups share a CU (one wave of each per SIMD).  Give the one whose LDS allocation starts at 0 a
    // higher static priority: it wins the matrix pipe whenever both want it, so the pair settles in anti-phase
    // (one contracting while the other runs its formal solution) instead of sharing the pipe and then idling it.
    if ((__builtin_amdgcn_s_getreg(0x3806) & 0xff) == 0) __builtin_amdgcn_s_setprio(SOS_STATIC_PRIO);
#endif
    // thread -> state row: t < 3N up-going (+mu), 3N <= t < 6N down-going; kk = c*N + jj in both halves
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool active = t < R6;
    const bool up = t < 3 * N;
    const int kk = active ? (up ? t : t - 3 * N) : 0;
    const int c = kk / N;
    const int jj = kk % N;                 // 0-based index of |direction|
    const int d = up ? jj : N + jj;
    const int rl = up ? kk : KH + kk;      // LDS row of this thread
    const int rsv = c * 2 * N + d;         // row in the order-1 vector tables (sv)
    const double mu = cx.mu[jj];
    const int recoff = c * W + N + (up ? (jj + 1) : -(jj + 1));
    const size_t mper = (size_t)2 * cx.rtph * cx.ks2h * 128;
    const int S1 = cx.smax + 1;

    {   // one workgroup = one bin (grid = nb): no bin loop, so per-bin constants are not kept live elsewhere
        const int b = blockIdx.x;
        const int nt = uniform_i32(bn.nt[b]);
        const int iborm = uniform_i32(bn.iborm[b]);
        const int jout = ZO ? uniform_i32(bn.jout ? bn.jout[b] : 0) : 0;
        const double zz = ZO ? uniform_f64(jout ? bn.zz[b] : 0.) : 0.;
        const int jlo = (ZO && jout) ? jout - 1 : -1, jhi = (ZO && jout) ? jout : -1;
        // shape guard (uniform): a malformed bin is flagged (norders = -1), never indexed out of bounds
        if (nt < 1 || nt >= LPB || nt >= bn.lp || iborm < 0 || iborm > cx.smax || jout < 0 || jout > nt) {
            if (t == 0) bn.norders[b] = -1;
            return;
        }
        const double *pf = bn.prof + (size_t)b * 3 * bn.lp;
        double *recb = bn.rec + (size_t)b * S1 * 3 * W;

        __syncthreads();
        if (t < N) { lga[t] = cx.ga[t]; lmu[t] = cx.mu[t]; }
        for (size_t i = t; i < (size_t)LPB * FS; i += 256) fld[i] = 0.;
        for (int i = t; i < LPB; i += 256) {
            const bool in = i <= nt;
            hh[i] = in ? pf[i] : 0.;
            xdel[i] = in ? pf[bn.lp + i] : 0.;
            ydel[i] = in ? pf[2 * bn.lp + i] : 0.;
        }
        __syncthreads();
        const double htot = uniform_f64(hh[nt]);
        int aer_l = 0;
        for (int i = t; i < LPB; i += 256) {
            if (i < nt) { const double dt = hh[i + 1] - hh[i]; dtau[i] = dt; idtau[i] = 1.0 / dt; }
            else { dtau[i] = 0.; idtau[i] = 0.; }
            ch[i] = (i <= nt) ? exp(-hh[i] / cx.mus) / 4. : 0.;                               // SOS_OS.F:837-839
            fco[i] = (i <= nt) ? (exp(-2. * htot / cx.mus) / 4.) * exp(hh[i] / cx.mus) : 0.;  // SOS_OS.F:3219,3278
            if (i <= nt && xdel[i] != 0.) aer_l = 1;
        }
        const int has_aer = uniform_i32(__syncthreads_or(aer_l));
        for (int i = t; i < nt * N; i += 256) att[i] = exp(-dtau[i / N] / cx.mu[i % N]);      // SOS_OS.F:2291,2335
        // bin-constant exponentials of the ground boundary terms (SOS_OS.F:979,985,1068-1077)
        const double e_sun = uniform_f64(exp(-htot / cx.mus));
        double e_mu = 0., e_lo = 0., e_hi = 0.;
        if (cx.imat_surf && active && up) {
            e_mu = exp(-htot / mu);
            e_lo = exp(-(htot - hh[0]) / mu);     // standard output: RIIOUT(0,K), SOS_OS.F:1068 (H(0) != 0)
            if (ZO && jout) { e_lo = exp(-(htot - hh[jlo]) / mu); e_hi = exp(-(htot - hh[jhi]) / mu); }
        }
        __syncthreads();

        // per-thread formal solution of its row, in place over the source held in fld (SOS_INTEGR_EPOPT,
        // SOS_OS.F:2279-2354).  bcv = value at the ground for up-going rows.  With t = exp(-dtau/|mu|) and the
        // source linear in tau on the layer, both directions reduce to the same three-term recurrence
        //     X_i = t X_n + p S_i + w S_n,   w = (1-t) |mu|/dtau - t,  p = (1-t) - w
        // (n = the level the ray comes from: i+1 for up-going, i-1 for down-going rows), algebraically the
        // reference update X t + (1-t)(a mu + b) -/+ a t dtau.  Up- and down-going rows share ONE instruction stream
        // (per-lane start level and stride), all addresses are running pointers: ~11 vector instructions per level.
        const double usign = (c == 2 && !up) ? -1. : 1.;     // U(-mu) is stored negated (see gemm_source)
        double xb, xlo = 0., xhi = 0.;
        auto scan_row = [&](double bcv) {
            if (!active) { xb = 0.; return; }
            const int i0 = up ? nt : 0;                 // level the ray starts from
            const int di = up ? -1 : 1;
            const int lay0 = up ? nt - 1 : 0;           // first layer crossed; the layer index moves with di
            double *sp = fld + (size_t)i0 * FS + rl;
            const ptrdiff_t sst = (ptrdiff_t)di * FS;
            const double *ap = att + lay0 * N + jj;
            const int ast = di * N;
            const double *dp = dtau + lay0;             // idtau = dtau + LPB
            double z = up ? bcv : 0.;
            double sn = *sp;                            // source at the level the ray comes from
            *sp = z;
            if (ZO) { if (i0 == jlo) xlo = z; if (i0 == jhi) xhi = z; }
            int lev = i0;
            auto level = [&]() {
                sp += sst;
                const double a = *ap, si = *sp, r = mu * dp[LPB];
                const double omt = 1.0 - a;
                const double w = omt * r - a;
                const double pq = omt - w;
                z = z * a + (pq * si + w * sn);
                *sp = z;
                sn = si;
                ap += ast; dp += di;
                if (ZO) { lev += di; if (lev == jlo) xlo = z; if (lev == jhi) xhi = z; }
            };
            int k = 0;
#pragma unroll 1
            for (; k + SU <= nt; k += SU) {             // exact trip counts: blocks of SU levels, then the tail
#pragma unroll
                for (int u = 0; u < SU; ++u) level();
            }
#pragma unroll 1
            for (; k < nt; ++k) level();
            xb = z;
            if (!up) gnd[c * N + jj] = z * usign;
        };

        double i4 = 0., i5 = 0.;
        double sign = -1.;
        int nord = 0;
#ifdef SOS_PROFILE_PHASES
        unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 order-1 fill, 1 scan, 2 gemm, 3 writeback, 4 reduce+tests, 5 ground_bc, 6 fourier, 7 init
#endif
        PH_T0();
        for (int s = 0; s <= iborm; ++s) {            // SOS_OS.F:872
            sign = -sign;
            const float *rs = cx.imat_surf ? cx.rsurf + (size_t)s * 9 * N * N : nullptr;
            // ground reflection of the down-going field of the previous order (SOS_OS.F:1166-1239)
            auto ground_bc = [&]() -> double {
                if (!(active && up)) return 0.;
                double v = 0., xr = 0.;
                if (c == 0 && cx.ro != 0. && s == 0) {
                    double lsol = 0.;
#pragma unroll 1
                    for (int j = 0; j < N; j++) lsol = lsol + lga[j] * gnd[j] * lmu[j];
                    lsol = 2 * lsol * cx.ro;
                    v = lsol; xr = lsol;
                }
                if (cx.imat_surf) {
                    double acc2 = 0.;
                    const float *r0 = rs + (size_t)(c * 3 + 0) * N * N + (size_t)jj * N;
                    const float *r1 = rs + (size_t)(c * 3 + 1) * N * N + (size_t)jj * N;
                    const float *r2 = rs + (size_t)(c * 3 + 2) * N * N + (size_t)jj * N;
#pragma unroll 1
                    for (int j = 0; j < N; j++) {
                        double q0 = r0[j], q1 = r1[j], q2 = r2[j];
                        if (!cx.ipolar) { q1 = 0.; q2 = 0.; if (c) q0 = 0.; }   // SOS_OS.F:928-941
                        acc2 = acc2 + lga[j] * (gnd[j] * q0 + gnd[N + j] * q1 + gnd[2 * N + j] * q2);
                    }
                    v = acc2 * (2 / mu) + xr;
                }
                if (cx.ifresnel == 1) {
                    const double f11 = cx.fres[jj], f12 = cx.fres[N + jj], f33 = cx.fres[2 * N + jj];
                    if (c == 0) v = v + f11 * gnd[jj] + f12 * gnd[N + jj];
                    else if (c == 1) v = v + f12 * gnd[jj] + f11 * gnd[N + jj];
                    else v = v + f33 * gnd[2 * N + jj];
                }
                return v;
            };

            // ---- scattering order 1 ------------------------------------------------------------
            const double *svp = cx.sv + (size_t)s * 4 * KP;
            if (active) {
                const double sva = svp[rsv], svr = svp[KP + rsv];
                const double sfa = svp[2 * KP + rsv], sfr = svp[3 * KP + rsv];
#pragma unroll 1
                for (int i = 0; i <= nt; i++) {
                    double v = ch[i] * (sva * xdel[i] + svr * ydel[i]);              // SOS_OS.F:2557-2559
                    if (cx.ifresnel == 1 && (up ? (i < nt) : (i >= 1)))
                        v = v + fco[i] * (sfa * xdel[i] + sfr * ydel[i]);            // SOS_OS.F:3280-3289
                    fld[(size_t)i * FS + rl] = v * usign;
                }
            }
            double bc = 0., dirterm = 0.;
            if (active && up) {                                                      // SOS_OS.F:970-992
                double xr = 0.;
                if (c == 0 && cx.ro != 0. && s == 0) { bc = cx.ro * cx.mus * e_sun; xr = bc; }
                if (cx.imat_surf) {
                    const double rr = e_sun / mu;
                    double r = rs[(size_t)(c * 3) * N * N + (size_t)jj * N + (cx.n0 - 1)];
                    if (!cx.ipolar && c) r = 0.;
                    bc = bc + r * rr;
                    dirterm = bc - xr;                                               // SOS_OS.F:1070-1072
                }
            }
            PH(0);
            scan_row(bc);
            PH(1);
            double rii = 0., riilo = 0., riihi = 0.;
            if (cx.imat_surf && active && up) {                                      // SOS_OS.F:1062-1084
                rii = e_mu * dirterm;
                riilo = e_lo * dirterm;
                if (ZO) riihi = e_hi * dirterm;
            }
            double i3 = xb, a1 = 0., d1 = xb, g1 = 0.;                               // SOS_OS.F:1094-1137
            double i3lo = ZO ? xlo : 0., dlo = i3lo, i3hi = ZO ? xhi : 0., dhi = i3hi;
            __syncthreads();
            bc = ground_bc();
            PH(5);

            // ---- scattering orders >= 2 --------------------------------------------------------
            int ig = 1, iglast = 1;
            for (;;) {
                ig = ig + 1;
                if (ig > cx.igmax) break;
                iglast = ig;
                // source function: dense FP64 contraction on the matrix cores (SOS_FSOURCE_ORDREIG)
                const int nchunk = BIG ? (nt + COLS) / COLS : 1;
#pragma unroll 1
                for (int chk = 0; chk < nchunk; chk++) {
                    const int l0 = chk * COLS;                 // first level of this chunk
                    v4d acc[2][RTWH][CT];
#pragma unroll
                    for (int sy = 0; sy < 2; sy++)
#pragma unroll
                        for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                            for (int ct = 0; ct < CT; ct++) acc[sy][rt][ct] = (v4d){0., 0., 0., 0.};
                    __syncthreads();
                    if (BIG) {                                 // stage the chunk: scratch -> LDS (16 B per lane, coalesced)
                        const int per_lev = KH;                // v2d per level (2*KH doubles)
                        for (int q = t; q < COLS * per_lev; q += 256) {
                            const int col = q / per_lev, r2 = q % per_lev;
                            v2d v = {0., 0.};
                            if (l0 + col <= nt) v = *reinterpret_cast<const v2d *>(fld + (size_t)(l0 + col) * FS + 2 * r2);
                            *reinterpret_cast<v2d *>(cbuf + (size_t)col * CS + 2 * r2) = v;
                        }
                        __syncthreads();
                    }
                    gemm_source<RTWH, CT>(acc, cx.mp_aer + (size_t)s * mper, has_aer != 0,
                                          cx.mp_vt + (size_t)(s <= 2 ? s : 0) * cx.ks2h * 128,
                                          cx.mp_uf + (size_t)(s <= 2 ? s : 0) * cx.rtph * 64, s <= 2 ? (s & 1) : -1,
                                          cx.ks2h, cx.rtph, 2 * N, cbuf, CS, KH, xdel + l0, ydel + l0, lane, wv);
                    __syncthreads();
                    PH(2);
                    // S+ = E^A + E^B, stored S- = E^A - E^B.  No lane predicates: pad rows (< KH) and pad columns of the
                    // accumulators are exact zeros (zero operator rows, zero field columns) and are stored as such.
                    {
                        double *wb = cbuf + (size_t)(lane & 15) * CS + (wv * RTWH) * 16 + (lane >> 4);
#pragma unroll
                        for (int ct = 0; ct < CT; ct++) {
                            double *wp = wb + (size_t)ct * 16 * CS, *wm = wp + KH;
#pragma unroll
                            for (int rt = 0; rt < RTWH; rt++) {
                                // rows of this tile inside the half system (KH is a multiple of 8, register e holds
                                // rows 4e..4e+3 of the tile): a wave-uniform count, no lane predicate
                                const int ne = (KH - (wv * RTWH + rt) * 16) >> 2;
#pragma unroll
                                for (int e = 0; e < 4; e++)
                                    if (e < ne) {
                                        const double ea = acc[0][rt][ct][e], eb = acc[1][rt][ct][e];
                                        wp[rt * 16 + 4 * e] = ea + eb;
                                        wm[rt * 16 + 4 * e] = ea - eb;
                                    }
                            }
                        }
                    }
                    __syncthreads();
                    if (BIG) {                                 // source chunk: LDS -> scratch
                        const int per_lev = KH;
                        for (int q = t; q < COLS * per_lev; q += 256) {
                            const int col = q / per_lev, r2 = q % per_lev;
                            if (l0 + col <= nt)
                                *reinterpret_cast<v2d *>(fld + (size_t)(l0 + col) * FS + 2 * r2) =
                                    *reinterpret_cast<const v2d *>(cbuf + (size_t)col * CS + 2 * r2);
                        }
                    }
                }
                __syncthreads();
                PH(3);
                scan_row(bc);                                                        // SOS_OS.F:1244
                __syncthreads();
                PH(1);
                g1 = xb;
                const double i3n = i3 + g1;
                int pm = 0;
                if (active) {
                    if (ig != 2 && conv_exceeds(a1, d1, g1, i3, cx.thr_cv)) pm |= 1;     // SOS_PARAM_CONV
                    const double ag = fabs(g1);
                    if (ag > cx.thr_val) pm |= 2;                                       // SOS_ARRET_DIFFUS_1
                    if (i3n != 0.0 && ag > cx.thr_sum * fabs(i3n)) pm |= 4;             // SOS_ARRET_DIFFUS_2
                }
                pm = block_or_bits(pm, reinterpret_cast<int *>(red));
                PH(4);
                bc = ground_bc();
                PH(5);
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
            // SOS_OS.F:1421-1439.  The record is built from I3OUT (minus RIIOUT at the output level), the stop
            // tests and fluxes from I3 (minus RII): the two differ by exp(H(0)/mu) on the direct term.
            double i3out0 = i3;
            if (cx.imat_surf && active && up) {
                i3out0 = i3 - riilo; i3 = i3 - rii;
                if (ZO) { i3lo = i3lo - riilo; i3hi = i3hi - riihi; }
            }

            if (s == 0) {                                                            // SOS_OS.F:1447-1456
                if (active && c == 0) i3s[d] = i3;
                __syncthreads();
                if (t == 0) {
                    double em = 0., ep = 0.;
                    for (int j = 0; j < N; j++) {
                        em = em + lmu[j] * lga[j] * i3s[N + j];
                        ep = ep + lmu[j] * lga[j] * i3s[j];
                    }
                    bn.flux[2 * b] = em * 2 / cx.mus;
                    bn.flux[2 * b + 1] = ep * 2 / cx.mus;
                }
            }
            const double coef = (s == 0) ? 1. : 2.;                                  // SOS_OS.F:1460-1473
            i4 = i4 + coef * i3;
            i5 = i5 + coef * i3 * sign;
            if (active) {                                                            // SOS_OS.F:1484-1534,1572
                const double outv = (ZO && jout) ? ((1 - zz) * i3lo + zz * i3hi) : i3out0;
                recb[(size_t)s * 3 * W + recoff] = outv * usign;
                if (up && jj == 0) recb[(size_t)s * 3 * W + c * W + N] = 0.;
            }
            if (t == 0) bn.iglast[(size_t)b * S1 + s] = iglast;
            nord = s + 1;
            int pf = 0;                                                              // SOS_ARRET_FOURIER
            if (active) {
                const double a3 = fabs(i3);
                if ((i4 != 0.0 && a3 > cx.thr_sf * fabs(i4)) || (i5 != 0.0 && a3 > cx.thr_sf * fabs(i5))) pf = 1;
            }
            pf = block_or_bits(pf, reinterpret_cast<int *>(red));
            PH(6);
            if (!pf) break;                                                          // SOS_OS.F:1585
        }
        // orders not run: zero records and counts
        for (int i = t + nord * 3 * W; i < S1 * 3 * W; i += 256) recb[i] = 0.;
        for (int i = t + nord; i < S1; i += 256) bn.iglast[(size_t)b * S1 + i] = 0;
        if (t == 0) bn.norders[b] = nord;
#ifdef SOS_PROFILE_PHASES
        if (bn.phase && lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&bn.phase[(size_t)b * 8 + k], ph_acc[k]);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// variant table
// ---------------------------------------------------------------------------------------------
static size_t lds_bytes_for(int n, int ct, bool big)
{
    const int cols = 16 * ct;
    const int kh = sos_round_up(3 * n, 8);
    size_t dbl = (size_t)cols * (2 * kh + 2) + 3 * n + 2 * n + 16 + 2 * n;
    if (!big) dbl += (size_t)cols * n + 7 * cols;
    return dbl * sizeof(double);
}

// returns 0 and the variant (rtw = row tiles per wave and system, ct = column tiles, big) or UNSUPPORTED
int sos_os_variant(int n, int nt_max, int *rtw, int *ct, size_t *lds_bytes, int *big)
{
    if (n < 1 || n > 85 || nt_max < 1 || nt_max > 1023) return SOSGPU_E_UNSUPPORTED;
    const int kh = sos_round_up(3 * n, 8);
    const int rth = (kh + 15) / 16;
    const int r = (rth + 3) / 4;
    int c = 0, b = 0;
    if (nt_max + 1 <= 32 && lds_bytes_for(n, 2, false) <= 160 * 1024) c = 2;
    else if (nt_max + 1 <= 64 && r <= 3 && lds_bytes_for(n, 4, false) <= 160 * 1024 && !getenv("SOSGPU_DEBUG_FORCE_BIG")) c = 4;
    else { c = 2; b = 1; }
    const size_t lb = lds_bytes_for(n, c, b);
    if (lb > 160 * 1024) return SOSGPU_E_UNSUPPORTED;
    if (rtw) *rtw = r;
    if (ct) *ct = c;
    if (lds_bytes) *lds_bytes = lb;
    if (big) *big = b;
    return 0;
}

size_t sos_os_scratch_doubles(int n, int lpb)
{
    const int kh = sos_round_up(3 * n, 8);
    return (size_t)lpb * (2 * kh) + (size_t)lpb * n + 7 * (size_t)lpb;
}

template <int RTWH, int CT, bool BIG, bool ZO>
static int launch_variant(const SosDev &cx, const SosBins &bn, size_t lds, hipStream_t st)
{
    auto kern = k_sos_os<RTWH, CT, BIG, ZO>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -2;
    const int grid = bn.nb;
    kern<<<grid, 256, lds, st>>>(cx, bn);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sos_os(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st)
{
    int rtw, ct, big;
    size_t lds;
    const int rc = sos_os_variant(cx.n, nt_max, &rtw, &ct, &lds, &big);
    if (rc) return rc;
    if (cx.rtph != 4 * rtw) return SOSGPU_E_UNSUPPORTED;
    if (big && (!bn.scratch || bn.lpb < nt_max + 1)) return SOSGPU_E_UNSUPPORTED;
    const int zo = bn.jout != nullptr;
#define V(R, C, B)                                                                        \
    if (rtw == R && ct == C && big == B)                                                  \
        return zo ? launch_variant<R, C, B, true>(cx, bn, lds, st) : launch_variant<R, C, B, false>(cx, bn, lds, st);
    V(1, 2, 0) V(2, 2, 0) V(3, 2, 0) V(4, 2, 0)
    V(1, 4, 0) V(2, 4, 0) V(3, 4, 0)
    V(1, 2, 1) V(2, 2, 1) V(3, 2, 1) V(4, 2, 1)
#undef V
    return SOSGPU_E_UNSUPPORTED;
}
