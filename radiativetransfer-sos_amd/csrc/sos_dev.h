// csrc/sos_dev.h -- device helpers shared by the successive-orders solver kernels (sos_os.hip: field resident in LDS;
// sos_stream.hip: field streamed through LDS in chunks of 32 levels): workgroup predicate reduction, the stop tests of
// SOS_OS.F:3377-3793 in predicate form, the parity-decomposed FP64 MFMA source contraction (SOS_FSOURCE_ORDREIG) and the
// blocked formal solution (SOS_INTEGR_EPOPT).
#pragma once
#include <type_traits>
#include "sos_common.h"

#define SOSGPU_E_UNSUPPORTED -3
// Parity sums X+ +- X- of the B operands: formed by every wave inside the contraction (8 vector sums per k-pair and wave), or
// once per step / chunk in LDS.  Measured (profiles/r02_flagship_instmix.txt): in the LDS-resident kernel the combine pass needs
// a workgroup barrier of its own and LOSES 2.1 % (194.1 vs 198.2 k bins/s) -- off; in the streamed kernel it rides in the
// fix-up pass that closes the up-going rows anyway, no barrier added: +0.5 % -- on.
#ifndef SOS_PRECOMBINE_LDS
#define SOS_PRECOMBINE_LDS 0
#endif
#ifndef SOS_PRECOMBINE_STREAM
#define SOS_PRECOMBINE_STREAM 1
#endif
#ifdef SOS_PROFILE_PHASES
// s_memrealtime: constant 100 MHz counter (s_memtime is NOT wall-clock on gfx950 when waves share a SIMD: it advances
// at 1/k of the shader clock with k MFMA-streaming waves per SIMD -- scripts/ubench_mfma_peak.hip)
#define PH_T0() unsigned long long ph_t = __builtin_amdgcn_s_memrealtime()
#define PH(k) do { unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); ph_acc[k] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define PH_T0() do {} while (0)
#define PH(k) do {} while (0)
#endif

// Global address = wave-uniform base (scalar registers) + a 32-bit per-lane byte offset.  The offset is re-hidden from the
// optimiser at every use: folded into a per-lane 64-bit pointer it would cost two or three vector instructions per access,
// kept apart the access uses the scalar-base addressing form and costs none (vector issue is the scarce resource here).
template <typename T>
__device__ __forceinline__ T *lane_ptr(T *ubase, unsigned &voff)
{
    asm volatile("" : "+v"(voff));
    // (pointer arithmetic, not integer: the address space of ubase must stay visible or a flat access is generated)
    return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(ubase)) + voff);
}

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

// bitwise OR over the workgroup of up to 3 predicate bits (red: 2 x NW ints of LDS, the two halves used alternately:
// `slot` flips on every call, so a wave that runs ahead into the next call writes the other half and ONE barrier per
// call is enough -- the barrier also publishes whatever the caller wrote to LDS/scratch before).  No FP64 work.
// (__syncthreads_or only returns a logical OR.)
template <int NW>
__device__ __forceinline__ int block_or_word(int w, int *red, int wv, int lane, int &slot)
{
    int *r = red + slot * NW;
    slot ^= 1;
    if (lane == 0) r[wv] = w;
    __syncthreads();
    int o = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) o |= r[i];
    return uniform_i32(o);
}
// The predicates are passed as such: a comparison leaves its result as a wave mask in scalar registers, and "any lane" of it
// is a scalar test -- packing them into per-lane bits first and unpacking them for the ballots cost a dozen vector instructions.
template <int NW>
__device__ __forceinline__ int block_or_bits(bool b1, bool b2, bool b4, int *red, int wv, int lane, int &slot)
{
    int w = 0;
    if (__ballot(b1)) w |= 1;
    if (__ballot(b2)) w |= 2;
    if (__ballot(b4)) w |= 4;
    return block_or_word<NW>(w, red, wv, lane, slot);
}
template <int NW>
__device__ __forceinline__ int block_or_bits(bool b1, int *red, int wv, int lane, int &slot)
{
    return block_or_word<NW>(__ballot(b1) ? 1 : 0, red, wv, lane, slot);
}

// Instruction budget.  FP64 vector instructions share the FP64 datapath with v_mfma_f64 on gfx950 (equal peak
// rates), and two workgroups share a CU, one wave of each per SIMD.  While one wave streams v_mfma_f64 (64 cycles
// each, back to back) its partner gets roughly one vector-issue slot per MFMA: measured with phase stamps, every
// non-MFMA vector instruction of the formal solution / write-back / tests costs 20-30 cycles in this regime
// WHATEVER its kind.  The non-contraction phases are therefore written for the LOWEST VECTOR INSTRUCTION COUNT
// (immediate offsets, wave-uniform control flow, no predicates, no divisions, no FP reductions), not for latency.
//
// Stop tests: "max_k |y_k| > thr" is the same decision as "exists k: |num_k| > thr |den_k|".
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

// Field storage convention: half-system order kk = c*N + (k-1) in both direction halves, and the U component of the
// down-going half is stored NEGATED (V- = -U(-mu)).  With that the parity combinations need no per-row sign:
//   X^A = X+ + X-,  X^B = X+ - X-,  S+ = E^A + E^B,  S- = E^A - E^B
// (for U: X^A_U = U+ - U-, stored S-_U = -(E^B - E^A)); the formal solution is linear with a zero boundary for
// down-going rows, so it maps a negated source to a negated field.  The sign is restored where U(-mu) leaves the
// field: ground values (gnd) and output records.
//
// Source contraction in the parity-decomposed form (sos_common.h): for both half systems
//   acc[sys] = XDEL o (M^sys X^sys)                                              (aerosol operator, dense)
// plus, for s <= 2, the molecular operator in its exact rank-4 form on the one half system it acts on:
//   acc[sr] += U (YDEL o (V^T X^sr))                                             (noyaux.hip k_pack_ray)
// A wave works on NA (1 or 2) row tiles tile0, tile0 + NW of BOTH systems x CT column tiles; bx = its B-fragment
// base (column lane&15, k-quad lane>>4) in the field (LDS, or the HBM scratch of the BIG variants) with level stride FS.
// Software pipeline: the A fragments (global/L2) and the raw B operands X+, X- (LDS) of k-pair m+1 are requested
// before the 16+ MFMAs of k-pair m are issued (two register sets each, used alternately, no copies), so neither the
// L2 nor the LDS round trip sits between two k-pairs.  The rank-4 projection of the molecular operator (RAY = half
// system it acts on, -1 = none) either comes for free (FOLD: its four rows are packed into padding rows of the dense
// operator) or rides in the same loop as 2 CT extra MFMAs per k-pair fed by one more prefetched 16-byte fragment.
template <int NA, int RAY, bool FOLD, int RTWH, int CT, int NW, int FS, int KHM, bool PIPE_B, bool PRE = false>
__device__ __forceinline__ void gemm_source(v4d (&acc)[2][RTWH][CT], const double *__restrict__ mp, bool do_aer,
                                            const double *__restrict__ vt, const double *__restrict__ uf,
                                            int ks2h, int rtph, const double *bx, const double *xdel,
                                            const double *ydel, int lane, int tile0, int prow, double *pcb)
{
    // fold: the projection rows V^T sit in padding rows prow..prow+3 of the packed aerosol operator (api.hip), so the
    // dense pass below already computes V^T X^sr in the accumulator quad of those rows; pcb = this lane's slot in the
    // (unused) padding rows of the LDS buffer through which the owning wave hands the projections to the others
    constexpr bool fold = FOLD;          // the caller selects it: RAY >= 0, aerosol present, prow >= 0
    struct BRaw { v2d xp[CT], xm[CT]; };
    auto load_b = [&](BRaw &b, int m) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            b.xp[ct] = *reinterpret_cast<const v2d *>(bx + ct * 16 * FS + 8 * m);
            b.xm[ct] = *reinterpret_cast<const v2d *>(bx + ct * 16 * FS + KHM + 8 * m);
        }
    };
    const v2d *vp = reinterpret_cast<const v2d *>(vt) + lane;
    v4d pr[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) pr[ct] = (v4d){0., 0., 0., 0.};
    if (do_aer) {
        const size_t rts = (size_t)ks2h * 64;               // v2d stride between row tiles
        const size_t sys_stride = (size_t)rtph * rts;       // v2d stride between the two systems
        // Operand addresses = wave-uniform byte base (scalar registers, advanced by scalar adds) + the lane's 32-bit offset:
        // the scalar-base form of global_load needs no vector address arithmetic at all (it was 12 of the 28 non-MFMA vector
        // instructions per pair of k-pairs).  `loff` is re-hidden from the optimiser at every use so that it is not folded
        // into a per-lane 64-bit pointer again.
        const char *ab = reinterpret_cast<const char *>(mp) + (size_t)tile0 * rts * 16;
        const char *vb = reinterpret_cast<const char *>(vt);
        unsigned loff = (unsigned)lane * 16u;
        struct AFrag { v2d a[2][NA]; v2d v; };
        auto load_a = [&](AFrag &f, int m) {
            // (the k-pair offset is hidden as well: derived from the previous call's bases it would need an immediate of
            // 4096, one more than the instruction can hold, and fall back to vector arithmetic)
            size_t mb = (size_t)m * 1024;
            asm volatile("" : "+v"(loff), "+s"(mb));
#pragma unroll
            for (int sy = 0; sy < 2; sy++)
#pragma unroll
                for (int rt = 0; rt < NA; rt++)
                    f.a[sy][rt] = *reinterpret_cast<const v2d *>(ab + (sy * sys_stride + (size_t)rt * NW * rts) * 16 + mb + loff);
            if (RAY >= 0 && !fold) f.v = *reinterpret_cast<const v2d *>(vb + mb + loff);
        };
        // FIRST: the first k-pair of the contraction starts from a zero accumulator given as the inline constant operand of
        // the MFMA, so the caller's zero fill of the 2 RTWH CT accumulators (32 vector moves per step) disappears
        auto mma = [&](const AFrag &f, const BRaw &b, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const v4d zero = {0., 0., 0., 0.};
            v2d ba[CT], bb[CT];
#pragma unroll
            // PRE: the buffer already holds X^A = X+ + X- where X+ used to be and X^B = X+ - X- where X- used to be (combined
            // once per step for all four waves instead of by every wave for itself: 8 sums per k-pair and wave less)
            for (int ct = 0; ct < CT; ct++) {
#ifdef SOS_EXP_NOADD                                       // timing experiment only (wrong physics): the contraction without its sums
                if (true) { ba[ct] = b.xp[ct]; bb[ct] = b.xm[ct]; }
#else
                if (PRE) { ba[ct] = b.xp[ct]; bb[ct] = b.xm[ct]; }
#endif
                else { ba[ct] = b.xp[ct] + b.xm[ct]; bb[ct] = b.xp[ct] - b.xm[ct]; }
            }
#pragma unroll
            for (int rt = 0; rt < NA; rt++)
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    acc[0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[0][rt].x, ba[ct].x, FIRST ? zero : acc[0][rt][ct], 0, 0, 0);
                    acc[1][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[1][rt].x, bb[ct].x, FIRST ? zero : acc[1][rt][ct], 0, 0, 0);
                }
#pragma unroll
            for (int rt = 0; rt < NA; rt++)
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    acc[0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[0][rt].y, ba[ct].y, acc[0][rt][ct], 0, 0, 0);
                    acc[1][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[1][rt].y, bb[ct].y, acc[1][rt][ct], 0, 0, 0);
                }
            if (RAY >= 0 && !fold) {
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    const v2d bq = RAY ? bb[ct] : ba[ct];
                    pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v.x, bq.x, pr[ct], 0, 0, 0);
                    pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v.y, bq.y, pr[ct], 0, 0, 0);
                }
            }
        };
        const std::true_type first;
        const std::false_type next;
        AFrag f0, f1;
        BRaw b0, b1;
        load_a(f0, 0);
        int m = 0;
        if (PIPE_B) {
            load_b(b0, 0);
            if (ks2h >= 2) {                                  // peeled first pair of k-pairs
                load_a(f1, 1);
                load_b(b1, 1);
                mma(f0, b0, first);
                if (2 < ks2h) { load_a(f0, 2); load_b(b0, 2); }
                mma(f1, b1, next);
                m = 2;
#pragma unroll 1
                for (; m + 1 < ks2h; m += 2) {
                    load_a(f1, m + 1);
                    load_b(b1, m + 1);
                    mma(f0, b0, next);
                    if (m + 2 < ks2h) { load_a(f0, m + 2); load_b(b0, m + 2); }
                    mma(f1, b1, next);
                }
                if (m < ks2h) mma(f0, b0, next);
            } else mma(f0, b0, first);
        } else {
            // four column tiles: one set of B registers (32 VGPRs) -- the second set costs more in spills than it hides
            if (ks2h >= 2) {
                load_a(f1, 1);
                load_b(b0, 0);
                mma(f0, b0, first);
                if (2 < ks2h) load_a(f0, 2);
                load_b(b0, 1);
                mma(f1, b0, next);
                m = 2;
#pragma unroll 1
                for (; m + 1 < ks2h; m += 2) {
                    load_a(f1, m + 1);
                    load_b(b0, m);
                    mma(f0, b0, next);
                    if (m + 2 < ks2h) load_a(f0, m + 2);
                    load_b(b0, m + 1);
                    mma(f1, b0, next);
                }
                if (m < ks2h) { load_b(b0, m); mma(f0, b0, next); }
            } else { load_b(b0, 0); mma(f0, b0, first); }
        }
        if (fold) {
            // accumulator register 3 of the tile holding prow is row prow + (lane>>4): the projection, in the B-operand
            // layout of the K = 4 expansion step.  The owner publishes it and clears it (those rows are padding).
            const int ptile = prow >> 4;
#pragma unroll
            for (int rt = 0; rt < NA; rt++)
                if (tile0 + rt * NW == ptile) {
#pragma unroll
                    for (int ct = 0; ct < CT; ct++) {
                        pcb[ct * 16 * FS] = acc[RAY > 0][rt][ct][3];
                        acc[RAY > 0][rt][ct][3] = 0.;
                    }
                }
        }
        // XDEL of the output level: every accumulator register of a lane belongs to one column.  Without a molecular term
        // (RAY < 0: Fourier orders > 2) the factor is applied by the write-back instead, fused into its sums
        // (write_back_source: 3 instead of 4 vector instructions per accumulator pair).
        if (RAY >= 0)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const double sc = xdel[ct * 16 + (lane & 15)];
#pragma unroll
            for (int sy = 0; sy < 2; sy++)
#pragma unroll
                for (int rt = 0; rt < NA; rt++) acc[sy][rt][ct] *= sc;
        }
    } else if (RAY >= 0) {
        // molecular atmosphere: only the projections pr = V^T X^sr (one 16-row tile, rows 0..3 used, per column tile)
        v2d v0 = vp[0], v1;
        BRaw b0, b1;
        load_b(b0, 0);
        auto prj = [&](const v2d &v, const BRaw &b) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const v2d bq = PRE ? (RAY ? b.xm[ct] : b.xp[ct]) : (RAY ? b.xp[ct] - b.xm[ct] : b.xp[ct] + b.xm[ct]);
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, bq.x, pr[ct], 0, 0, 0);
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, bq.y, pr[ct], 0, 0, 0);
            }
        };
        int m = 0;
#pragma unroll 1
        for (; m + 1 < ks2h; m += 2) {
            v1 = vp[(size_t)(m + 1) * 64];
            load_b(b1, m + 1);
            prj(v0, b0);
            if (m + 2 < ks2h) { v0 = vp[(size_t)(m + 2) * 64]; load_b(b0, m + 2); }
            prj(v1, b1);
        }
        if (m < ks2h) prj(v0, b0);
    }
    if (RAY >= 0) {
        if (fold) __syncthreads();             // every wave of the workgroup passes exactly one barrier here (see the call site)
        // register 0 of pr holds row (lane>>4) in 0..3, column lane&15: exactly the B-operand layout of one
        // K = 4 step, so the expansion U * (YDEL o pr) needs no data movement
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const double q = (fold ? pcb[ct * 16 * FS] : pr[ct][0]) * ydel[ct * 16 + (lane & 15)];
#pragma unroll
            for (int rt = 0; rt < NA; rt++) {
                const double u = uf[(size_t)(tile0 + rt * NW) * 64 + lane];
                acc[RAY > 0][rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, q, acc[RAY > 0][rt][ct], 0, 0, 0);
            }
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------------
// SHARED-TILE form of the contraction for direction counts whose half system has MORE row tiles than waves but fewer than two
// per wave (four waves: 5 or 6 row tiles, N = 22 ... 32 -- the reference's default of 24 Gauss angles gives N = 25, 5 tiles).
// In gemm_source wave 0 (and 1) then carries a second tile alone: per k-pair it issues 16 MFMAs while the others issue 8 and
// wait at the barrier.  Here every wave owns ONE row tile (tile0 = its index, both half systems, all column tiles) and the
// R = rtph - NW left-over tiles are cut into (half system, column tile) units: wave w takes unit (usy, uct) = (w >> 1, w & 1)
// of every left-over tile -- 8 + 2 R MFMAs per k-pair for every wave (10 instead of 16 on the critical wave at R = 1).
// The unit's accumulators hold E^USY of one column tile only; S+ = E^A + E^B, S- = E^A - E^B of a left-over tile are formed
// through the field buffer: the partials go to the pad rows [16 rtph, 16 rtph + 16 R) of the half USY (rows no operand,
// sweep or projection slot ever touches: write_back_shared), and after a barrier combine_shared adds them up in place.
// Everything else -- operand layout, software pipeline, the folded molecular projection, XDEL / YDEL scaling -- is gemm_source's.
template <int RAY, bool FOLD, int CT, int NW, int FS, int KHM, bool PRE = false>
__device__ __forceinline__ void gemm_source_split(v4d (&acc)[2][1][CT], v4d (&accs)[2], const int R, const int usy, const int uct,
                                                  const double *__restrict__ mp, bool do_aer, const double *__restrict__ vt,
                                                  const double *__restrict__ uf, int ks2h, int rtph, const double *bx,
                                                  const double *xdel, const double *ydel, int lane, int tile0, int prow, double *pcb)
{
    // (usy, uct are wave-uniform run-time values: as template arguments they would quadruple the already large kernels.  The
    //  unit's B operand X^usy of column tile uct is read from the field a second time instead of being selected out of the
    //  registers of the own tile's operands: one or two more LDS reads per k-pair.)
    static_assert(CT == 2 && NW == 4, "units are (half system, column tile) of four waves");
    constexpr bool fold = FOLD;
    struct BRaw { v2d xp[CT], xm[CT], up, um; };
    const double *bxu = bx + uct * 16 * FS + (PRE && usy ? KHM : 0);
    const double sgn = usy ? -1. : 1.;
    auto load_b = [&](BRaw &b, int m) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            b.xp[ct] = *reinterpret_cast<const v2d *>(bx + ct * 16 * FS + 8 * m);
            b.xm[ct] = *reinterpret_cast<const v2d *>(bx + ct * 16 * FS + KHM + 8 * m);
        }
        b.up = *reinterpret_cast<const v2d *>(bxu + 8 * m);
        if (!PRE) b.um = *reinterpret_cast<const v2d *>(bxu + KHM + 8 * m);
    };
    const v2d *vp = reinterpret_cast<const v2d *>(vt) + lane;
    const bool ray_unit = RAY >= 0 && usy == (RAY > 0 ? 1 : 0);      // this wave's unit lies in the half system the molecular operator acts on
    v4d pr[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) pr[ct] = (v4d){0., 0., 0., 0.};
    if (do_aer) {
        const size_t rts = (size_t)ks2h * 64;               // v2d stride between row tiles
        const size_t sys_stride = (size_t)rtph * rts;       // v2d stride between the two systems
        const char *ab = reinterpret_cast<const char *>(mp) + (size_t)tile0 * rts * 16;
        const char *sb = reinterpret_cast<const char *>(mp) + ((size_t)usy * sys_stride + (size_t)NW * rts) * 16;   // left-over tiles
        const char *vb = reinterpret_cast<const char *>(vt);
        unsigned loff = (unsigned)lane * 16u;
        struct AFrag { v2d a[2]; v2d as[2]; v2d v; };
        auto load_a = [&](AFrag &f, int m) {
            size_t mb = (size_t)m * 1024;
            asm volatile("" : "+v"(loff), "+s"(mb));
#pragma unroll
            for (int sy = 0; sy < 2; sy++) f.a[sy] = *reinterpret_cast<const v2d *>(ab + (sy * sys_stride) * 16 + mb + loff);
            f.as[0] = *reinterpret_cast<const v2d *>(sb + mb + loff);
            if (R > 1) f.as[1] = *reinterpret_cast<const v2d *>(sb + rts * 16 + mb + loff);
            if (RAY >= 0 && !fold) f.v = *reinterpret_cast<const v2d *>(vb + mb + loff);
        };
        auto mma = [&](const AFrag &f, const BRaw &b, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const v4d zero = {0., 0., 0., 0.};
            v2d ba[CT], bb[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                if (PRE) { ba[ct] = b.xp[ct]; bb[ct] = b.xm[ct]; }
                else { ba[ct] = b.xp[ct] + b.xm[ct]; bb[ct] = b.xp[ct] - b.xm[ct]; }
            }
            v2d bu;
            if (PRE) bu = b.up;
            else { bu.x = __builtin_fma(sgn, b.um.x, b.up.x); bu.y = __builtin_fma(sgn, b.um.y, b.up.y); }   // X+ +- X-, exact
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                acc[0][0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[0].x, ba[ct].x, FIRST ? zero : acc[0][0][ct], 0, 0, 0);
                acc[1][0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[1].x, bb[ct].x, FIRST ? zero : acc[1][0][ct], 0, 0, 0);
            }
            accs[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.as[0].x, bu.x, FIRST ? zero : accs[0], 0, 0, 0);
            if (R > 1) accs[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.as[1].x, bu.x, FIRST ? zero : accs[1], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                acc[0][0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[0].y, ba[ct].y, acc[0][0][ct], 0, 0, 0);
                acc[1][0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[1].y, bb[ct].y, acc[1][0][ct], 0, 0, 0);
            }
            accs[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.as[0].y, bu.y, accs[0], 0, 0, 0);
            if (R > 1) accs[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.as[1].y, bu.y, accs[1], 0, 0, 0);
            if (RAY >= 0 && !fold) {
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    const v2d bq = RAY ? bb[ct] : ba[ct];
                    pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v.x, bq.x, pr[ct], 0, 0, 0);
                    pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v.y, bq.y, pr[ct], 0, 0, 0);
                }
            }
        };
        const std::true_type first;
        const std::false_type next;
        AFrag f0, f1;
        BRaw b0, b1;
        load_a(f0, 0);
        int m = 0;
        load_b(b0, 0);
        if (ks2h >= 2) {                                  // peeled first pair of k-pairs (as in gemm_source, PIPE_B form)
            load_a(f1, 1);
            load_b(b1, 1);
            mma(f0, b0, first);
            if (2 < ks2h) { load_a(f0, 2); load_b(b0, 2); }
            mma(f1, b1, next);
            m = 2;
#pragma unroll 1
            for (; m + 1 < ks2h; m += 2) {
                load_a(f1, m + 1);
                load_b(b1, m + 1);
                mma(f0, b0, next);
                if (m + 2 < ks2h) { load_a(f0, m + 2); load_b(b0, m + 2); }
                mma(f1, b1, next);
            }
            if (m < ks2h) mma(f0, b0, next);
        } else mma(f0, b0, first);
        if (fold) {
            // the projection rows sit in register 3 of the tile holding prow: this wave's own tile (all columns), or a left-over
            // tile -- then the two waves holding the half system the molecular operator acts on publish their column each
            const int ptile = prow >> 4;
            if (tile0 == ptile) {
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    pcb[ct * 16 * FS] = acc[RAY > 0][0][ct][3];
                    acc[RAY > 0][0][ct][3] = 0.;
                }
            }
            if (ray_unit) {
                if (ptile == NW) { pcb[uct * 16 * FS] = accs[0][3]; accs[0][3] = 0.; }
                if (R > 1 && ptile == NW + 1) { pcb[uct * 16 * FS] = accs[1][3]; accs[1][3] = 0.; }
            }
        }
        if (RAY >= 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const double sc = xdel[ct * 16 + (lane & 15)];
                acc[0][0][ct] *= sc; acc[1][0][ct] *= sc;
            }
            const double su = xdel[uct * 16 + (lane & 15)];
            accs[0] *= su;
            if (R > 1) accs[1] *= su;
        }
    } else if (RAY >= 0) {
        // molecular atmosphere: only the projections pr = V^T X^sr
        v2d v0 = vp[0], v1;
        BRaw b0, b1;
        load_b(b0, 0);
        auto prj = [&](const v2d &v, const BRaw &b) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const v2d bq = PRE ? (RAY ? b.xm[ct] : b.xp[ct]) : (RAY ? b.xp[ct] - b.xm[ct] : b.xp[ct] + b.xm[ct]);
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, bq.x, pr[ct], 0, 0, 0);
                pr[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, bq.y, pr[ct], 0, 0, 0);
            }
        };
        int m = 0;
#pragma unroll 1
        for (; m + 1 < ks2h; m += 2) {
            v1 = vp[(size_t)(m + 1) * 64];
            load_b(b1, m + 1);
            prj(v0, b0);
            if (m + 2 < ks2h) { v0 = vp[(size_t)(m + 2) * 64]; load_b(b0, m + 2); }
            prj(v1, b1);
        }
        if (m < ks2h) prj(v0, b0);
    }
    if (RAY >= 0) {
        if (fold) __syncthreads();             // every wave of the workgroup passes exactly one barrier here (see the call site)
        double qu = 0.;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const double q = (fold ? pcb[ct * 16 * FS] : pr[ct][0]) * ydel[ct * 16 + (lane & 15)];
            const double u = uf[(size_t)tile0 * 64 + lane];
            acc[RAY > 0][0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(u, q, acc[RAY > 0][0][ct], 0, 0, 0);
            if (ct == uct) qu = q;
        }
        if (ray_unit) {
            accs[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[(size_t)NW * 64 + lane], qu, accs[0], 0, 0, 0);
            if (R > 1) accs[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[(size_t)(NW + 1) * 64 + lane], qu, accs[1], 0, 0, 0);
        }
    }
}

// Partials of the left-over tiles -> pad rows of the field buffer: unit (USY, UCT) of left-over tile r goes to rows
// 16 rtph + 16 r ... of the half USY, columns of column tile UCT (lane: column lane&15, rows 4e + (lane>>4)).
// xdel != nullptr: the accumulators are still unscaled (s > 2).
template <int FS, int KHM>
__device__ __forceinline__ void write_back_shared(const v4d (&accs)[2], int R, int usy, int uct, double *cbuf, int lane, int rtph,
                                                  const double *xdel)
{
    double *wb = cbuf + (size_t)(uct * 16 + (lane & 15)) * FS + usy * KHM + 16 * rtph + (lane >> 4);
    const double sc = xdel ? xdel[uct * 16 + (lane & 15)] : 1.0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        wb[4 * e] = xdel ? sc * accs[0][e] : accs[0][e];
        if (R > 1) wb[16 + 4 * e] = xdel ? sc * accs[1][e] : accs[1][e];
    }
}

// S+ = E^A + E^B, S- = E^A - E^B of the left-over tiles (rows 16 NW ... 16 rtph of both halves) from the partials in the pad
// rows; NTH threads, one pass, consecutive threads on consecutive rows.  A barrier before (partials written) and after.
template <int NW, int FS, int KHM, int COLS>
__device__ __forceinline__ void combine_shared(double *cbuf, int t, int nth, int rtph)
{
    const int nrow = 16 * (rtph - NW);
    for (int i = t; i < nrow * COLS; i += nth) {
        const int col = i / nrow, row = i - col * nrow;
        double *c = cbuf + (size_t)col * FS;
        const double a = c[16 * rtph + row], b = c[KHM + 16 * rtph + row];
        c[16 * NW + row] = a + b;
        c[KHM + 16 * NW + row] = a - b;
    }
}

// Parity combination of the whole field buffer in place, once per step (flagship: between the stop tests and the contraction;
// one more workgroup barrier after it): wave w takes levels w, w + NW, ..., a lane two adjacent rows of both halves.
template <int NW, int FS, int KHM, int LEVELS>
__device__ __forceinline__ void combine_field(double *fld, int lane, int wv)
{
    static_assert(KHM % 16 == 0, "rows per half");
#pragma unroll
    for (int l = 0; l < LEVELS / NW; l++) {
        double *row = fld + (size_t)(wv + l * NW) * FS;
#pragma unroll
        for (int r = 0; r < KHM; r += 128) {
            if (KHM - r >= 128 || 2 * lane < KHM - r) {
                v2d *pp = reinterpret_cast<v2d *>(row + r + 2 * lane), *pm = reinterpret_cast<v2d *>(row + KHM + r + 2 * lane);
                const v2d xp = *pp, xm = *pm;
                *pp = xp + xm; *pm = xp - xm;
            }
        }
    }
}

// Write-back of the source function: S+ = E^A + E^B, stored S- = E^A - E^B (see the storage convention above), one 16 x 16
// accumulator tile per (row tile, column tile): lane (column lane&15, row quad lane>>4) stores rows 4e + (lane>>4) of its tile.
// No lane predicates: pad rows (< KH) and pad columns of the accumulators are exact zeros (zero operator rows, zero field
// columns) and are stored as such.  Row tiles beyond KH are skipped by WAVE-UNIFORM tests; a wave whose tiles all lie inside
// the half system (every wave at N = 41) takes the straight-line form -- the per-store tests of the general form cost two
// vector instructions and a branch each.
// xdel != nullptr (wave-uniform): the accumulators are still unscaled (gemm_source with RAY < 0 and an aerosol operator);
// S+ = x ea + x eb, S- = x ea - x eb as one product and two fused multiply-adds.
template <int RTWH, int CT, int NW, int FS, int KHM>
__device__ __forceinline__ void write_back_source(const v4d (&acc)[2][RTWH][CT], double *cbuf, int lane, int wv, int kh,
                                                  const double *xdel)
{
    double *wb = cbuf + (lane & 15) * FS + (lane >> 4) + wv * 16;
    if ((wv + (RTWH - 1) * NW) * 16 + 16 <= kh) {
        if (xdel) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const double sc = xdel[ct * 16 + (lane & 15)];
#pragma unroll
                for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const double ea = acc[0][rt][ct][e], eb = acc[1][rt][ct][e];
                        const double p = sc * ea;
                        wb[ct * 16 * FS + rt * NW * 16 + 4 * e] = __builtin_fma(sc, eb, p);
                        wb[ct * 16 * FS + KHM + rt * NW * 16 + 4 * e] = __builtin_fma(-sc, eb, p);
                    }
            }
            return;
        }
#pragma unroll
        for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const double ea = acc[0][rt][ct][e], eb = acc[1][rt][ct][e];
                    wb[ct * 16 * FS + rt * NW * 16 + 4 * e] = ea + eb;
                    wb[ct * 16 * FS + KHM + rt * NW * 16 + 4 * e] = ea - eb;
                }
    } else {
        double sc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) sc[ct] = xdel ? xdel[ct * 16 + (lane & 15)] : 1.0;
#pragma unroll
        for (int rt = 0; rt < RTWH; rt++) {
            // rows of this tile inside the half system (kh is a multiple of 8, register e holds rows 4e..4e+3 of the tile)
            const int ne = (kh - (wv + rt * NW) * 16) >> 2;
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (e < ne) {
                        const double ea = sc[ct] * acc[0][rt][ct][e], eb = sc[ct] * acc[1][rt][ct][e];
                        wb[ct * 16 * FS + rt * NW * 16 + 4 * e] = ea + eb;
                        wb[ct * 16 * FS + KHM + rt * NW * 16 + 4 * e] = ea - eb;
                    }
        }
    }
}

// Ground reflection of the down-going field of the previous scattering order by a BRDF/BPDF surface (SOS_OS.F:1194-1220):
//   X_a(NT, k) = (2/mu_k) sum_j w_j sum_b X_b(NT, -j) R_ab(j, k)            (+ the Lambertian term on I for s = 0)
// a 3N x 3N matrix-vector product per scattering order.  As per-thread dot products over REAL*4 matrices in L2 it cost as much
// as half a source contraction (latency of 3N loads per row); here it is a small matrix-core product against the operator G_s
// (SosDev::mp_gnd) on v_mfma_f64_4x4x4f64 -- four independent 4 x 4 x 4 blocks per instruction, 16 cycles: the four blocks
// are four row quads of one 16-row tile, the vector is broadcast to the four columns, so an instruction does 64 useful
// multiply-adds in a quarter of the time the 16 x 16 x 4 form needs for the same 64.  Operand layout (probed on gfx950,
// scripts/probe_mfma4x4.hip): A[blk][i][k] in lane i + 4 blk + 16 k, B[blk][k][j] in lane j + 4 blk + 16 k,
// D[blk][i][j] in lane j + 4 blk + 16 i.  Packed operator (api.hip k_pack_ground):
//   gp[((tile * KS2H + m) * 64 + lane) * 2 + e] = G[tile*16 + 4 ((lane>>2)&3) + (lane&3)][8 m + 4 e + (lane>>4)]
// Wave w does row tiles {w, w + NW}; gndk = ground vector in half-system order (LDS); the lanes of column 0 publish bcv.
template <int NA, int NW>
__device__ __forceinline__ void ground_mfma(const double *__restrict__ gp, int ks2h, const double *gndk, double *bcv, int lane,
                                            int tile0)
{
    double acc[NA];
#pragma unroll
    for (int rt = 0; rt < NA; rt++) acc[rt] = 0.;
    const size_t rts = (size_t)ks2h * 64;
    // scalar byte base + 32-bit lane offset, as in gemm_source: scalar-base loads, no vector address arithmetic
    const char *ab = reinterpret_cast<const char *>(gp) + (size_t)tile0 * rts * 16;
    unsigned loff = (unsigned)lane * 16u;
    const double *bp = gndk + (lane >> 4);
#ifndef SOS_GROUND_MB
#define SOS_GROUND_MB 5
#endif
    constexpr int MB = SOS_GROUND_MB;         // k-pairs requested together
#pragma unroll 1
    for (int m = 0; m < ks2h; m += MB) {
        v2d a[MB][NA];
        double b0[MB], b1[MB];
#pragma unroll
        for (int u = 0; u < MB; u++) {
            const int mm = min(m + u, ks2h - 1);
            size_t mb = (size_t)mm * 1024;
            asm volatile("" : "+v"(loff), "+s"(mb));
#pragma unroll
            for (int rt = 0; rt < NA; rt++) a[u][rt] = *reinterpret_cast<const v2d *>(ab + (size_t)rt * NW * rts * 16 + mb + loff);
            b0[u] = bp[8 * mm]; b1[u] = bp[8 * mm + 4];
        }
#pragma unroll
        for (int u = 0; u < MB; u++)
            if (m + u < ks2h) {                                               // wave-uniform
#pragma unroll
                for (int rt = 0; rt < NA; rt++) {
                    acc[rt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][rt].x, b0[u], acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u][rt].y, b1[u], acc[rt], 0, 0, 0);
                }
            }
    }
    if ((lane & 3) == 0) {
#pragma unroll
        for (int rt = 0; rt < NA; rt++) bcv[(tile0 + rt * NW) * 16 + ((lane >> 2) & 3) * 4 + (lane >> 4)] = acc[rt];
    }
}

// Capacity constants of a variant (shared by the kernel and the host-side sizing below).
//   KHM = rows per direction half (3N <= KHM), FS = level stride of the field, NS = row stride of the attenuation table
__host__ __device__ constexpr int sos_khm(int nw, int rtwh) { return 16 * nw * rtwh; }
__host__ __device__ constexpr int sos_fs(int nw, int rtwh) { return 2 * sos_khm(nw, rtwh) + 2; }
__host__ __device__ constexpr int sos_ns(int nw, int rtwh) { return (sos_khm(nw, rtwh) / 3 + 1) & ~1; }

// Formal solution of one row (SOS_INTEGR_EPOPT, SOS_OS.F:2279-2354).  With t = exp(-dtau/|mu|) and the source linear
// in tau on the layer, both directions reduce to the same three-term recurrence
//     X_i = t X_n + (p S_i + w S_n),   w = (1-t) |mu|/dtau - t,  p = (1-t) - w
// evaluated as  X_i = X_n + (1-t)(S_i - X_n) + w (S_n - S_i)  with (1-t) from the per-bin table: 6 FP64 instructions per
// level (3 of them on the dependent chain) instead of 7
// (n = the level the ray comes from: i+1 for up-going, i-1 for down-going rows), algebraically the reference update
// X t + (1-t)(a mu + b) -/+ a t dtau.  A block of U levels is processed in three passes so that neither the LDS
// round trip nor the FP64 latency sits between two levels: (1) all operands of the block are loaded (the stores of
// the block come last, so the loads cannot be held back by possible aliasing), (2) the U independent coefficient /
// source combinations run with full instruction-level parallelism, (3) only one FMA per level is on the dependent
// chain.  DI = -1 sweeps from level nt down to 0 (up-going rows), DI = +1 from 0 to nt; all offsets are immediates.
// O1 = true: the source is the single-scattering source of the direct beam (SOS_FSOURCE_ORDRE1, SOS_OS.F:2557-2559, with the
// Fresnel-reflected beam of SOS_FSOURCE_DIFF_FRESNEL1 :3280-3289 when `fres`), formed on the fly from the per-level factors
//   S1_i = sva (ch_i XDEL_i) + svr (ch_i YDEL_i) [+ sfa (fco_i XDEL_i) + sfr (fco_i YDEL_i)]
// (lx -> ch XDEL at the level the ray has reached; the three other level vectors follow at multiples of lstr), instead of
// being written to the field and read back.
struct Order1 { double sva, svr, sfa, sfr; bool fres; };

// LDS pointers of the formal solution.  An explicit address space keeps them 32-bit, and lds_pin hides a freshly computed
// base from the optimiser: DS instructions take an UNSIGNED 16-bit immediate offset, and for the up-going sweep the compiler
// otherwise keeps the high end of a block as its induction pointer and pays one vector add per access for the negative offsets.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ void lds_pin(lds_f64 *&p)
{
    unsigned a = (unsigned)(unsigned long)p;
    asm volatile("" : "+v"(a));
    p = (lds_f64 *)(unsigned long)a;
}
__device__ __forceinline__ void lds_pin(const lds_f64 *&p)
{
    unsigned a = (unsigned)(unsigned long)p;
    asm volatile("" : "+v"(a));
    p = (const lds_f64 *)(unsigned long)a;
}

template <int DI, int U, int FS, int NS, bool O1>
__device__ __forceinline__ void scan_block(lds_f64 *&q, const lds_f64 *&qa, const lds_f64 *&qd, double mu, double &z, double &sn,
                                           const lds_f64 *&lx, int lstr, const Order1 &o1)
{
    // The up-going sweep (DI < 0) first moves its base pointers to the low end of the block and indexes upwards from there.
    // LV(u): offset, in levels, of the level reached after u + 1 steps; LA(u): of the layer crossed by step u + 1.
    if (DI < 0) {
        q -= U * FS; qa -= (U - 1) * NS; qd -= (U - 1);
        if (O1) lx -= U;
        lds_pin(q); lds_pin(qa);
    }
#define LV(u) (DI < 0 ? (U - 1 - (u)) : ((u) + 1))
#define LA(u) (DI < 0 ? (U - 1 - (u)) : (u))
    double av[U], sv[U], cv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { av[u] = qa[LA(u) * NS]; cv[u] = qd[LA(u)]; }
    if (O1) {
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u] = o1.sva * lx[LV(u)] + o1.svr * lx[lstr + LV(u)];
        if (o1.fres) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                sv[u] = sv[u] + (o1.sfa * lx[2 * lstr + LV(u)] + o1.sfr * lx[3 * lstr + LV(u)]);
        }
        if (DI > 0) lx += U;
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u] = q[LV(u) * FS];
    }
    double dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                       // av = 1 - t (table), w = (1-t)(|mu|/dtau + 1) - 1
        cv[u] = av[u] * (mu * cv[u] + 1.0) - 1.0;       // w
        dv[u] = (u ? sv[u - 1] : sn) - sv[u];           // S_n - S_i
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { z = cv[u] * dv[u] + (z + av[u] * (sv[u] - z)); cv[u] = z; }
#pragma unroll
    for (int u = 0; u < U; ++u) q[LV(u) * FS] = cv[u];
    sn = sv[U - 1];
    if (DI > 0) { q += U * FS; qa += U * NS; qd += U; }
    else { qa -= NS; qd -= 1; }
#undef LV
#undef LA
}

// Down-going block of the field-in-HBM variant: the source of U consecutive levels is read from the LDS chunk the
// contraction has just written (qs), the field goes straight to the scratch (q); same arithmetic as scan_block<+1>.
template <int U, int FS, int NS>
__device__ __forceinline__ void scan_block_split(double *&q, const double *&qs, const double *&qa, const double *&qd, double mu,
                                                 double &z, double &sn)
{
    double av[U], sv[U], cv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { av[u] = qa[u * NS]; cv[u] = qd[u]; sv[u] = qs[u * FS]; }
    double dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                       // av = 1 - t (table), w = (1-t)(|mu|/dtau + 1) - 1
        cv[u] = av[u] * (mu * cv[u] + 1.0) - 1.0;       // w
        dv[u] = (u ? sv[u - 1] : sn) - sv[u];           // S_n - S_i
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { z = cv[u] * dv[u] + (z + av[u] * (sv[u] - z)); cv[u] = z; }
#pragma unroll
    for (int u = 0; u < U; ++u) q[u * FS] = cv[u];
    sn = sv[U - 1];
    q += U * FS; qs += U * FS; qa += U * NS; qd += U;
}

