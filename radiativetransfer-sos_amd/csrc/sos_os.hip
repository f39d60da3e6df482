// csrc/sos_os.hip -- fused successive-orders-of-scattering solver for a batch of CKD bins (gfx950).
//
// Replaces, for every bin of a wavelength at once, the reference call chain
//   SOS_PROC.F:3459-3594 (bin loop) -> SOS.F:554 -> SOS_OS (src/SOS_OS.F:303-1674)
// with its leaves SOS_FSOURCE_ORDRE1 (:2431), SOS_FSOURCE_ORDREIG (:2663), SOS_INTEGR_EPOPT (:2222),
// SOS_FSOURCE_DIFF_FRESNEL1 (:3106), the stop tests (:3377,:3497,:3585,:3709) and SOS_AJOUT_QUEUE (:3871).
//
// MI355X mapping (one workgroup of NW waves = one bin, all Fourier orders s and scattering orders ig):
//  * the radiance field of the current scattering order, 2 x 3N rows x NT+1 levels, lives in LDS for the whole
//    solve, laid out [level][row] with a COMPILE-TIME level stride FS = 2 KHM + 2 doubles (KHM = 16 NW RTWH rows
//    per direction half, the capacity of the variant), rows [0,KHM) = X(+mu), [KHM,2KHM) = X(-mu);
//  * the order-ig source is the parity-decomposed dense FP64 contraction (two 3N x 3N half systems, see
//    sos_common.h) on v_mfma_f64_16x16x4_f64: wave w owns row tiles {w, w+NW} of BOTH systems x CT column tiles
//    of accumulators; the operator streams from L2/MALL in pre-packed A-fragment order (1 KiB contiguous per wave
//    load), the B fragments X+ +- X- are formed from LDS (one ds_read_b128 per lane per pair of k-steps);
//  * the source overwrites the field in LDS, then one thread per (Stokes, direction) row runs the layer-by-layer
//    formal solution (SOS_INTEGR_EPOPT) in place; waves [0,NW/2) hold the up-going rows, waves [NW/2,NW) the
//    down-going rows, so the direction of the sweep is wave-uniform and every LDS address of a block of levels is
//    a compile-time immediate offset from one base register;
//  * stop tests are predicate bits OR-ed over the workgroup (no FP reductions, no divisions); the geometric-series
//    tail, the Fourier accumulation and the Fourier stop are per-thread register state.
// HBM traffic per bin: 3(NT+1) doubles in, (F*3*(2N+1) + small) doubles out; everything else is on chip
// or L2/MALL-resident operator reads shared by all bins.
#include <atomic>
#include <cstdlib>
#include "sos_dev.h"
#include "kernels.h"

// SOS_MULTI (set by sos_os_multi.hip, which includes this file): the same kernels under other names, reading the wavelength
// context of each bin from a device table (SosBins::ctxs) -- one launch then covers the bins of many wavelengths.
#ifdef SOS_MULTI
#define k_sos_os k_sos_os_multi
#define launch_sos_os launch_sos_os_multi
#endif

// NW   : waves per workgroup (4: N <= 42, 8: N <= 85); waves [0,NW/2) hold up-going rows, the rest down-going rows
// RTWH : row tiles per wave and half system (tile = wave + rt*NW)
// CT   : column tiles (16 levels each) held in LDS at once
// The whole field (NT+1 <= 16*CT levels) lives in LDS; larger level grids (reference profiles have NT = 100..600,
// SOS.h:202,229) run in sos_stream.hip, which streams the field through LDS chunk by chunk.
// SURF = true: BRDF/BPDF reflection matrices (IMAT_SURF = 1, SOS_OS.F:912-925); a template argument because its per-row
//             matrix pointers and direct-beam terms otherwise stay live across the contraction (30 spilled VGPRs, -2.6 %)
// ZO = true : output at an intermediate altitude (ZOUT != -1, SOS_OS.F:1511-1534) -- two extra levels per row are
//             read back from the field after every formal solution.
// Register bound: 256 architectural VGPRs (two workgroups per CU for NW = 4 when the LDS allows it).  The 512-register
// form (256 VGPRs + 256 AGPRs, one workgroup per CU) of the 4-wave CT = 4 variants was measured slower in round 1 (60.2k vs
// 63.6k bins/s at N = 41, NT = 60) and one instantiation gave wrong down-going rows then.  Re-examined in round 2 with
// -DSOS_WIDE_REGS (below; builds <4,2,4,.> in that form and routes the <8,1,4> cases to it): all 85 parity and fuzz
// tests pass, zout_n25_nt60 included (gpurun log wide_regs.log, profiles/r02_wide_regs.txt) -- the round-1 failure does not
// reproduce on the current code (its write-back and ZOUT read-back were rewritten since), so there is no evidence of an
// ordering bug hidden by register allocation in the shipped variants; the form stays unused because it is slower.
#ifdef SOS_WIDE_REGS                                       // diagnostic build: 512-register form of the 4-wave CT = 4 variants
#define SOS_MIN_WG(NW, CT) (((NW) == 4 && (CT) != 4) ? 2 : 1)
#else
#define SOS_MIN_WG(NW, CT) (((NW) == 4) ? 2 : 1)
#endif
// The one-row-tile variants of up to 32 levels (N <= 21: 41 KB of LDS per workgroup) run THREE workgroups per CU (round 3): 168
// registers per lane instead of 256 cost 19-61 spilled registers, the third wave per SIMD hides more of the formal solution's
// latency than that costs -- N = 13 / 17 / 21 at NT = 30: 700 / 602 / 482 -> 820 / 687 / 541 k bins/s (profiles/r03_n_sweep.txt).
// (The same for N = 22 ... 26 -- a five-tile field layout as in sos_stream.hip, 51 KB per workgroup, three workgroups per CU, wave 0
// carrying the fifth tile -- was built and measured against the shared-tile form on two workgroups: 375.7 vs 371.1 k bins/s at
// N = 25, 335.1 vs 336.5 at N = 26, with 132-185 spilled registers; a wash, not kept.)
#ifdef SOS_NO_3WG
#define SOS_MIN_WG_R(NW, RTWH, CT) SOS_MIN_WG(NW, CT)
#else
#define SOS_MIN_WG_R(NW, RTWH, CT) (((NW) == 4 && (RTWH) == 1 && (CT) == 2) ? 3 : SOS_MIN_WG(NW, CT))
#endif
template <int NW, int RTWH, int CT, bool ZO, bool SURF, bool SPLIT>
__global__ __launch_bounds__(64 * NW, SOS_MIN_WG_R(NW, RTWH, CT)) void k_sos_os(const SosDev cx_arg, const SosBins bn)
{
    SOS_BIND_CTX(cx, cx_arg, bn);
    extern __shared__ double smem[];
    constexpr int NTH = 64 * NW, HW = NW / 2;
    constexpr int COLS = 16 * CT;
    constexpr int KHM = sos_khm(NW, RTWH), FS = sos_fs(NW, RTWH), NS = sos_ns(NW, RTWH);
    const int N = cx.n, KP = cx.kp, KH = cx.kh, W = cx.w;
    constexpr int LPB = COLS;              // level capacity of the field storage
    double *cbuf = smem;                   // [COLS][FS]  LDS: the field itself
    double *gnd = cbuf + COLS * FS;        // down-going field at the ground, order ig-1: [3][NS] by (component, direction), or --
                                           // SURF -- [KHM] in half-system order, the B operand of ground_mfma (3 NS + 2 >= KHM)
    double *red = gnd + 3 * NS + 2;        // [16]
    double *i3s = red + 16;                // [2][NS] I3 of the I rows (flux integrals)
    double *lga = i3s + 2 * NS;            // [NS] Gauss weights, [NS] mu (LDS copies for the ground-reflection sums)
    double *lmu = lga + NS;
    double *bcv = i3s;                     // SURF: [KHM] reflected field of ground_mfma, over i3s | lga | lmu (4 NS >= KHM), which
                                           // the SURF variants do not use while a Fourier order runs
    double *sbase = lmu + NS;
    double *fld = cbuf;                    // [LPB][FS]   field / source, [level][+mu rows | -mu rows]
    double *att = sbase;                   // [LPB][NS] exp(-dtau_i/mu_j), layer i = levels i..i+1
    double *idtau = att + (size_t)LPB * NS;  // [LPB] each:
    double *xdel = idtau + LPB;
    double *ydel = xdel + LPB;
    double *cxd = ydel + LPB;              // order-1 level factors: exp(-h/mus)/4 * XDEL, * YDEL
    double *cyd = cxd + LPB;
    double *fxd = cyd + LPB;               // Fresnel order-1 level factors (hold dtau / h during the set-up)
    double *fyd = fxd + LPB;

    // thread -> state row: waves [0,HW) up-going (+mu), waves [HW,NW) down-going; kk = c*N + jj in both halves
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool up = wv < HW;               // wave-uniform
    const int kk0 = up ? t : t - 64 * HW;
    const bool active = kk0 < 3 * N;
    const int kk = active ? kk0 : 0;       // half-system position of this thread's row
    const int cj = cx.rowmap[kk];          // -> component and direction (weighted directions first, sos_common.h)
    const int c = cj / N;
    const int jj = cj % N;                 // 0-based index of |direction|
    const int d = up ? jj : N + jj;
    const int rl = up ? kk : KHM + kk;     // field row of this thread
    const int rsv = c * 2 * N + d;         // row in the order-1 vector tables (sv)
    const double mu = cx.mu[jj];
    const int recoff = c * W + N + (up ? (jj + 1) : -(jj + 1));
    const size_t mper = (size_t)2 * cx.rtph * cx.ks2h * 128;
    const int S1 = cx.smax + 1;
    // shared-tile form of the contraction (sos_dev.h gemm_source_split): four waves, five or six row tiles (N = 22 ... 32).
    // A template argument, chosen by the launcher from the context's tile count (sos_split_applies): an instantiation of its own,
    // so that the kernels of the other direction counts are compiled exactly as without it.
    constexpr bool SPLIT_OK = SPLIT;
    static_assert(!SPLIT || (NW == 4 && RTWH == 2 && CT == 2), "shared-tile form: four waves, two column tiles");
    constexpr bool split = SPLIT;
    // contraction tiles of this wave: tile wv, and tile wv + NW for the larger N of a variant (wave-uniform)
    const bool tile_a = wv * 16 < KH;
    const bool tile_b = RTWH > 1 && (wv + NW) * 16 < KH;
    const double *bx = cbuf + (lane & 15) * FS + 2 * (lane >> 4);
    double *pcb = cbuf + (lane & 15) * FS + (cx.prow >= 0 ? cx.prow : 0) + (lane >> 4);    // see gemm_source (fold)

    {   // one workgroup = one bin (grid = nb): no bin loop, so per-bin constants are not kept live elsewhere
        const int b = SOS_BIN_INDEX(bn);
        const int nt = uniform_i32(bn.nt[b]);
        const int iborm = uniform_i32(bn.iborm[b]);
        const int jout = ZO ? uniform_i32(bn.jout ? bn.jout[b] : 0) : 0;
        const double zz = ZO ? uniform_f64(jout ? bn.zz[b] : 0.) : 0.;
        const int jlo = (ZO && jout) ? jout - 1 : 0, jhi = (ZO && jout) ? jout : 0;
        // shape guard (uniform): a malformed bin is flagged (norders = -1), never indexed out of bounds
        if (nt < 1 || nt >= LPB || nt >= bn.lp || iborm < 0 || iborm > cx.smax || jout < 0 || jout > nt) {
            if (t == 0) { bn.norders[b] = -1; bn.flux[2 * b] = 0.; bn.flux[2 * b + 1] = 0.; }
            for (int i = t; i < cx.smax + 1; i += NTH) bn.iglast[(size_t)b * (cx.smax + 1) + i] = 0;
            return;
        }
        const double *pf = bn.prof + (size_t)b * 3 * bn.lp;
        double *recb = bn.rec + (size_t)b * S1 * 3 * W;

        __syncthreads();
        if (!SURF && t < N) { lga[t] = cx.ga[t]; lmu[t] = cx.mu[t]; }
        if (SURF) for (int i = t; i < 3 * NS + 2; i += NTH) gnd[i] = 0.;
        for (size_t i = t; i < (size_t)LPB * FS; i += NTH) fld[i] = 0.;
        double *hh = fyd, *dtau = fxd;         // set-up only: these two slots end up holding the Fresnel factors
        for (int i = t; i < LPB; i += NTH) {
            const bool in = i <= nt;
            hh[i] = in ? pf[i] : 0.;
            xdel[i] = in ? pf[bn.lp + i] : 0.;
            ydel[i] = in ? pf[2 * bn.lp + i] : 0.;
        }
        __syncthreads();
        const double htot = uniform_f64(hh[nt]);
        const double h0 = uniform_f64(hh[0]);
        const double hlo = uniform_f64(hh[jlo]), hhi = uniform_f64(hh[jhi]);
        int aer_l = 0;
        for (int i = t; i < LPB; i += NTH) {
            const double dt = (i < nt) ? hh[i + 1] - hh[i] : 0.;
            dtau[i] = dt;
            idtau[i] = (i < nt) ? 1.0 / dt : 0.;
            const double ch = (i <= nt) ? exp(-hh[i] / cx.mus) / 4. : 0.;                               // SOS_OS.F:837-839
            cxd[i] = ch * xdel[i]; cyd[i] = ch * ydel[i];
            if (i <= nt && xdel[i] != 0.) aer_l = 1;
        }
        const int has_aer = uniform_i32(__syncthreads_or(aer_l));
        for (int i = t; i < nt * N; i += NTH) att[(i / N) * NS + i % N] = 1.0 - exp(-dtau[i / N] / cx.mu[i % N]);   // SOS_OS.F:2291,2335
        __syncthreads();
        for (int i = t; i < LPB; i += NTH) {   // own elements only: h_i -> fco_i XDEL_i | fco_i YDEL_i
            const double fco = (i <= nt) ? (exp(-2. * htot / cx.mus) / 4.) * exp(hh[i] / cx.mus) : 0.;  // SOS_OS.F:3219,3278
            fxd[i] = fco * xdel[i]; fyd[i] = fco * ydel[i];
        }
        // bin-constant exponentials of the ground boundary terms (SOS_OS.F:979,985,1068-1077)
        const double e_sun = uniform_f64(exp(-htot / cx.mus));
        __syncthreads();

        // per-thread formal solution of its row, in place over the source held in fld.  bcv = value at the ground for
        // up-going rows.  The sweep direction DI is a template argument (wave-uniform): blocks of 8/4/2/1 levels with
        // all field / attenuation / 1/dtau addresses as immediate offsets from three base registers.
        const double usign = (c == 2 && !up) ? -1. : 1.;     // U(-mu) is stored negated (see gemm_source)
        double xb = 0., xlo = 0., xhi = 0.;
        Order1 o1 = {0., 0., 0., 0., false};
        auto scan_dir = [&](auto dir_tag, auto o1_tag, double bcv) {
            constexpr int DI = decltype(dir_tag)::value;
            constexpr bool O1 = decltype(o1_tag)::value;
            // q  -> field at the level the ray has reached, qa -> attenuation of the next layer, qd -> its 1/dtau
            const int i0 = DI < 0 ? nt : 0;
            lds_f64 *q = (lds_f64 *)(fld + (size_t)i0 * FS + rl);
            const lds_f64 *qa = (const lds_f64 *)(att + (size_t)(DI < 0 ? nt - 1 : 0) * NS + jj);
            const lds_f64 *qd = (const lds_f64 *)(idtau + (DI < 0 ? nt - 1 : 0));
            const lds_f64 *lx = (const lds_f64 *)(cxd + i0);
            double z = bcv;
            // source at the level the ray comes from (no reflected-beam term there: SOS_OS.F:3280 excludes it)
            double sn = O1 ? o1.sva * lx[0] + o1.svr * lx[LPB] : *q;
            *q = z;
            int k = 0;
#pragma unroll 1
            for (; k + 8 <= nt; k += 8) scan_block<DI, 8, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, LPB, o1);
            if (nt & 4) scan_block<DI, 4, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, LPB, o1);
            if (nt & 2) scan_block<DI, 2, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, LPB, o1);
            if (nt & 1) scan_block<DI, 1, FS, NS, O1>(q, qa, qd, mu, z, sn, lx, LPB, o1);
            xb = z;
        };
        auto scan_row = [&](auto o1_tag, double bcv) {
            constexpr bool O1 = decltype(o1_tag)::value;
            if (up) {
                if (active) scan_dir(std::integral_constant<int, -1>(), o1_tag, bcv);
            } else if (active) {
                scan_dir(std::integral_constant<int, 1>(), o1_tag, 0.);
                gnd[SURF ? kk : c * NS + jj] = xb * usign;
            }
            if (ZO && jout && active) { xlo = fld[(size_t)jlo * FS + rl]; xhi = fld[(size_t)jhi * FS + rl]; }
        };

        double i4 = 0., i5 = 0.;
        double sign = -1.;
        int red_slot = 0;
        int nord = 0;
#ifdef SOS_PROFILE_PHASES
        unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 order-1 fill, 1 scan, 2 gemm, 3 writeback, 4 reduce+tests, 5 ground_bc, 6 fourier, 7 init
#endif
        PH_T0();
        for (int s = 0; s <= iborm; ++s) {            // SOS_OS.F:872
            sign = -sign;
            // ground reflection of the down-going field of the previous order (SOS_OS.F:1166-1239); every thread of the
            // workgroup calls it (the SURF form is a workgroup-wide matrix-core product with one barrier)
            const double *gop = SURF ? cx.mp_gnd + (size_t)s * cx.rtph * cx.ks2h * 128 : nullptr;
            const int gstep = (jj < cx.nwgt || !SURF) ? cx.nwgt : N - cx.nwgt;    // distance of (b, jj) from (c, jj) in gnd (SURF)
            auto ground_bc = [&]() -> double {
                double v = 0.;
                if (SURF) {
                    // BRDF/BPDF matrices (SOS_OS.F:1194-1220) and the Lambertian part, folded into G_0: see ground_mfma
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
                    unsigned jj8 = (unsigned)jj * 8u;                 // scalar base + lane offset: no vector address arithmetic
                    const double f11 = *lane_ptr(cx.fres, jj8), f12 = *lane_ptr(cx.fres + N, jj8), f33 = *lane_ptr(cx.fres + 2 * N, jj8);
                    const double *g0 = SURF ? gnd + kk - c * gstep : gnd + jj;      // (I, jj); Q and U follow at gs, 2 gs
                    const int gs = SURF ? gstep : NS;
                    if (c == 0) v = v + f11 * g0[0] + f12 * g0[gs];
                    else if (c == 1) v = v + f12 * g0[0] + f11 * g0[gs];
                    else v = v + f33 * g0[2 * gs];
                }
                return v;
            };

            // ---- scattering order 1: row factors of the single-scattering source (SOS_FSOURCE_ORDRE1); the source itself
            // is formed inside the first formal solution (scan_block<O1>); they carry the storage sign of U(-mu)
            const double *svp = cx.sv + (size_t)s * 4 * KP;
            if (active) {
                o1.sva = svp[rsv] * usign; o1.svr = svp[KP + rsv] * usign;
                o1.fres = cx.ifresnel == 1;
                if (o1.fres) { o1.sfa = svp[2 * KP + rsv] * usign; o1.sfr = svp[3 * KP + rsv] * usign; }
            }
            // direct solar beam reflected by the surface into this (up-going) row: Lambert part xr, BRDF/BPDF part
            // dirterm = R_a1(N0,k) exp(-tau/mus)/mu_k (SOS_OS.F:970-992)
            auto direct_beam = [&](double &xr) -> double {
                double v = 0.;
                xr = 0.;
                if (c == 0 && cx.ro != 0. && s == 0) { v = cx.ro * cx.mus * e_sun; xr = v; }
                if (SURF) v = v + cx.rdir[((size_t)s * 3 + c) * N + jj] * (e_sun / mu);      // polarisation cut applied at packing
                return v;
            };
            double bc = 0., dirterm = 0.;          // dirterm is only kept (SURF variants) for the end of the order
            if (active && up) { double xr; bc = direct_beam(xr); if (SURF) dirterm = bc - xr; }  // SOS_OS.F:1070-1072
            PH(0);

            // ---- scattering orders: formal solution of the current source, stop tests, next source ----------
            double i3 = 0., a1 = 0., d1 = 0., g1 = 0.;
            double i3lo = 0., dlo = 0., i3hi = 0., dhi = 0.;
            int ig = 1, iglast = 1;
            for (;;) {
                if (ig == 1) scan_row(std::true_type(), bc);                         // SOS_OS.F:1025
                else scan_row(std::false_type(), bc);                                // SOS_OS.F:1244
                if (ig == 1) {                                                       // SOS_OS.F:1094-1137
                    __syncthreads();
                    PH(1);
                    i3 = xb; a1 = 0.; d1 = xb; g1 = 0.;
                    if (ZO) { i3lo = xlo; dlo = xlo; i3hi = xhi; dhi = xhi; }
                    bc = ground_bc();
                    PH(5);
                } else {
                    PH(1);                                   // formal solution without its barrier wait (that goes to 4)
                    g1 = xb;
                    const double i3n = i3 + g1;
                    const double ag = fabs(g1);
                    const bool p1 = active && ig != 2 && conv_exceeds(a1, d1, g1, i3, cx.thr_cv);       // SOS_PARAM_CONV
                    const bool p2 = active && ag > cx.thr_val;                                          // SOS_ARRET_DIFFUS_1
                    const bool p4 = active && i3n != 0.0 && ag > cx.thr_sum * fabs(i3n);                // SOS_ARRET_DIFFUS_2
                    // the barrier inside also ends the formal solution: field and ground values are complete after it
                    const int pm = block_or_bits<NW>(p1, p2, p4, reinterpret_cast<int *>(red), wv, lane, red_slot);
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
                ig = ig + 1;
                if (ig > cx.igmax) break;
                iglast = ig;
                // source function of order ig: dense FP64 contraction on the matrix cores (SOS_FSOURCE_ORDREIG)
                if (SOS_PRECOMBINE_LDS) {
                    // X^A, X^B formed once for the four waves (sos_dev.h combine_field); the barrier ends the pass
                    combine_field<NW, FS, KHM, CT * 16>(fld, lane, wv);
                    __syncthreads();
                }
                {
                    v4d acc[2][RTWH][CT];       // (zero for a bin without aerosol operator; the dense pass starts from the MFMA zero operand)
#pragma unroll
                    for (int sy = 0; sy < 2; sy++)
#pragma unroll
                        for (int rt = 0; rt < RTWH; rt++)
#pragma unroll
                            for (int ct = 0; ct < CT; ct++) acc[sy][rt][ct] = (v4d){0., 0., 0., 0.};
                    const double *mpa = cx.mp_aer + (size_t)s * mper;
                    const double *vtp = cx.mp_vt + (size_t)(s <= 2 ? s : 0) * cx.ks2h * 128;
                    const double *ufp = cx.mp_uf + (size_t)(s <= 2 ? s : 0) * cx.rtph * 64;
                    // RAY: half system the molecular operator acts on (A for even s, B for odd s), none for s > 2
                    const bool fold = s <= 2 && has_aer && cx.prow >= 0;       // projection rows ride in the dense pass
                    auto contract = [&](auto na_tag) {
                        constexpr int NA = decltype(na_tag)::value;
#define SOS_GEMM(RAYV, FOLDV)                                                                                          \
    gemm_source<NA, RAYV, FOLDV, RTWH, CT, NW, FS, KHM, (CT < 4), (SOS_PRECOMBINE_LDS != 0)>(acc, mpa, has_aer != 0, vtp, ufp, cx.ks2h, cx.rtph, bx,        \
                                                        xdel, ydel, lane, wv, cx.prow, pcb)
                        if (s > 2) SOS_GEMM(-1, false);
                        else if (s & 1) { if (fold) SOS_GEMM(1, true); else SOS_GEMM(1, false); }
                        else { if (fold) SOS_GEMM(0, true); else SOS_GEMM(0, false); }
#undef SOS_GEMM
                    };
                    if constexpr (SPLIT_OK) {
                        if (split) {
                            // more row tiles than waves, fewer than two per wave (N = 22 ... 32): one own tile per wave, the
                            // left-over tiles shared out by (half system, column tile) -- sos_dev.h gemm_source_split
                            v4d own[2][1][CT], accs[2] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}};
#pragma unroll
                            for (int sy = 0; sy < 2; sy++)
#pragma unroll
                                for (int ct = 0; ct < CT; ct++) own[sy][0][ct] = (v4d){0., 0., 0., 0.};
                            const int R = cx.rtph - NW;
                            {
#define SOS_GEMM_S(RAYV, FOLDV)                                                                                        \
    gemm_source_split<RAYV, FOLDV, CT, NW, FS, KHM, (SOS_PRECOMBINE_LDS != 0)>(own, accs, R, wv >> 1, wv & 1, mpa, has_aer != 0, vtp, ufp, cx.ks2h, \
                                                               cx.rtph, bx, xdel, ydel, lane, wv, cx.prow, pcb)
                                if (s > 2) SOS_GEMM_S(-1, false);
                                else if (s & 1) { if (fold) SOS_GEMM_S(1, true); else SOS_GEMM_S(1, false); }
                                else { if (fold) SOS_GEMM_S(0, true); else SOS_GEMM_S(0, false); }
#undef SOS_GEMM_S
                            }
                            __syncthreads();         // every wave has read the field
                            PH(2);
                            const double *xd = (s > 2 && has_aer) ? xdel : nullptr;
                            write_back_source<1, CT, NW, FS, KHM>(own, cbuf, lane, wv, KH, xd);
                            write_back_shared<FS, KHM>(accs, R, wv >> 1, wv & 1, cbuf, lane, cx.rtph, xd);
                            __syncthreads();
                            combine_shared<NW, FS, KHM, CT * 16>(cbuf, t, NTH, cx.rtph);
                            __syncthreads();
                        }
                    }
                    if (!(SPLIT_OK && split)) {
                        if (tile_b) contract(std::integral_constant<int, RTWH>());
                        else if (tile_a) contract(std::integral_constant<int, 1>());
                        else if (fold) __syncthreads();                        // the barrier of the folded projection
                        __syncthreads();             // every wave has read the field
                        PH(2);
                        write_back_source<RTWH, CT, NW, FS, KHM>(acc, cbuf, lane, wv, KH, (s > 2 && has_aer) ? xdel : nullptr);
                        __syncthreads();
                    }
                }
                PH(3);
            }
            // SOS_OS.F:1421-1439.  The record is built from I3OUT (minus RIIOUT at the output level), the stop
            // tests and fluxes from I3 (minus RII): the two differ by exp(H(0)/mu) on the direct term.
            double i3out0 = i3;
            if (SURF && active && up) {                                      // SOS_OS.F:1062-1084
                const double rii = exp(-htot / mu) * dirterm;
                // standard output: RIIOUT(0,K), SOS_OS.F:1068 (H(0) != 0)
                const double riilo = exp(-(htot - ((ZO && jout) ? hlo : h0)) / mu) * dirterm;
                i3out0 = i3 - riilo; i3 = i3 - rii;
                if (ZO) {
                    const double riihi = (jout ? exp(-(htot - hhi) / mu) : 0.) * dirterm;
                    i3lo = i3lo - riilo; i3hi = i3hi - riihi;
                }
            }

            if (s == 0) {                                                            // SOS_OS.F:1447-1456
                if (active && c == 0) i3s[up ? jj : NS + jj] = i3;
                __syncthreads();
                if (t == 0) {
                    double em = 0., ep = 0.;
                    for (int j = 0; j < N; j++) {          // (mu, weights from global: the SURF variants keep no LDS copy)
                        em = em + cx.mu[j] * cx.ga[j] * i3s[NS + j];
                        ep = ep + cx.mu[j] * cx.ga[j] * i3s[j];
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
            const double a3 = fabs(i3);                                                             // SOS_ARRET_FOURIER
            const bool pf = active && ((i4 != 0.0 && a3 > cx.thr_sf * fabs(i4)) || (i5 != 0.0 && a3 > cx.thr_sf * fabs(i5)));
            const int pf2 = block_or_bits<NW>(pf, reinterpret_cast<int *>(red), wv, lane, red_slot);
            PH(6);
            if (!pf2) break;                                                         // SOS_OS.F:1585
        }
        // orders not run: their records are left unwritten (sosgpu_aggregate reads norders and skips them, like the
        // shorter FICOS file of the reference, SOS_AGGREGATE.F:357-413); only the counts are cleared
        for (int i = t + nord; i < S1; i += NTH) bn.iglast[(size_t)b * S1 + i] = 0;
        if (t == 0) bn.norders[b] = nord;
#ifdef SOS_PROFILE_PHASES
        if (bn.phase && lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&bn.phase[(size_t)b * 8 + k], ph_acc[k]);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// variant table
// ---------------------------------------------------------------------------------------------
#ifndef SOS_MULTI
static size_t lds_bytes_for(int nw, int rtw, int ct, bool big)
{
    const int cols = 16 * ct, fs = sos_fs(nw, rtw), ns = sos_ns(nw, rtw);
    size_t dbl = (size_t)cols * fs + 3 * ns + 2 + 2 * ns + 16 + 2 * ns;
    if (!big) dbl += (size_t)cols * ns + 7 * cols;
    return dbl * sizeof(double);
}

// Variant of a problem size: waves per workgroup, row tiles per wave, column tiles per chunk, field in LDS or in HBM.
//   NT+1 <= 32 : field in LDS, two column tiles, 4 waves (two workgroups per CU) -- the flagship
//   NT+1 <= 64 : field in LDS, four column tiles; for 21 < N <= 42 EIGHT waves with one row tile each, so that the
//                accumulators stay at 64 VGPRs and the CU still hosts two waves per SIMD although the 64-level field
//                leaves room for one workgroup only (68.6k vs 61.3k bins/s for <4,2,4> at N = 41, NT = 60)
//   NT+1 >  64 : field in the HBM scratch, two column tiles per contraction pass, two workgroups per CU (the 8-wave
//                four-tile form was slower there: 21.8k vs 26.5k bins/s at NT = 100)
//   N > 42     : eight waves, two row tiles each, two column tiles (LDS holds 32 levels of the 6N x 8 B rows)
static void sos_os_shape(int n, int nt_max, int *nw, int *rtw, int *ct, int *big)
{
    const int kh = sos_round_up(3 * n, 8);
    if (kh > 128) { *nw = 8; *rtw = 2; *ct = 2; *big = nt_max + 1 > 32; return; }
    if (nt_max + 1 <= 32 || nt_max + 1 > 64) { *nw = 4; *rtw = kh <= 64 ? 1 : 2; *ct = 2; *big = nt_max + 1 > 64; return; }
    *nw = kh <= 64 ? 4 : 8; *rtw = 1; *ct = 4; *big = 0;
}

// returns 0 and the variant (nw waves, rtw row tiles per wave and system, ct column tiles, big) or UNSUPPORTED
int sos_os_variant(int n, int nt_max, int *nw, int *rtw, int *ct, size_t *lds_bytes, int *big)
{
    if (n < 1 || n > 85 || nt_max < 1 || nt_max > 1023) return SOSGPU_E_UNSUPPORTED;
    int w, r, c, b;
    sos_os_shape(n, nt_max, &w, &r, &c, &b);
    const size_t lb = lds_bytes_for(w, r, c, b);
    if (lb > 160 * 1024) return SOSGPU_E_UNSUPPORTED;
    if (nw) *nw = w;
    if (rtw) *rtw = r;
    if (ct) *ct = c;
    if (lds_bytes) *lds_bytes = lb;
    if (big) *big = b;
    return 0;
}
#endif

// five or six row tiles on four waves, and room for the partials in the pad rows of the field (sos_dev.h gemm_source_split)
static bool sos_split_applies(const SosDev &cx, int nw, int rtw, int ct)
{
#ifdef SOS_NO_SPLIT
    return false;
#else
    return nw == 4 && rtw == 2 && ct == 2 && cx.rtph > nw && cx.rtph <= nw + 2 && 16 * cx.rtph + 16 * (cx.rtph - nw) <= sos_khm(nw, rtw);
#endif
}

template <int NW, int RTWH, int CT, bool ZO, bool SURF, bool SPLIT = false>
static int launch_variant(const SosDev &cx, const SosBins &bn, size_t lds, hipStream_t st, int *hip_err)
{
    if constexpr (!SPLIT && NW == 4 && RTWH == 2 && CT == 2) {
        if (sos_split_applies(cx, NW, RTWH, CT)) return launch_variant<NW, RTWH, CT, ZO, SURF, true>(cx, bn, lds, st, hip_err);
    }
    auto kern = k_sos_os<NW, RTWH, CT, ZO, SURF, SPLIT>;
#ifdef SOS_PROFILE_PHASES
    if (const char *e = getenv("SOSGPU_DEBUG_LDS_PAD")) lds += (size_t)atoi(e);   // diagnostic builds: force 1 workgroup per CU
#endif
    // the dynamic-LDS limit of a kernel is set once per device and size (the call costs tens of microseconds: with few bins per
    // wavelength the host launch path is what bounds a hyperspectral loop, scripts/spectrum_bench.py)
    // (atomics: host threads of run_sos.sos_proc_many launch concurrently; a lost update only repeats the call)
    static std::atomic<size_t> configured[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && (dev < 0 || dev >= 16 || configured[dev].load(std::memory_order_acquire) < lds)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && dev >= 0 && dev < 16) configured[dev].store(lds, std::memory_order_release);
    }
    if (e == hipSuccess) {
        kern<<<bn.nb, 64 * NW, lds, st>>>(cx, bn);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { if (hip_err) *hip_err = (int)e; return -2; }
    return 0;
}

int launch_sos_os(const SosDev &cx, const SosBins &bn, int nt_max, hipStream_t st, int *hip_err)
{
    int nw, rtw, ct, big;
    size_t lds;
    const int rc = sos_os_variant(cx.n, nt_max, &nw, &rtw, &ct, &lds, &big);
    if (rc) return rc;
    if (cx.kh > sos_khm(nw, rtw) || cx.rtph * 16 < cx.kh) return SOSGPU_E_UNSUPPORTED;
    if (big) return SOSGPU_E_UNSUPPORTED;                  // streamed variant: launch_sos_stream (sos_stream.hip)
    const int zo = bn.jout != nullptr;
#ifdef SOS_WIDE_REGS
    if (nw == 8 && rtw == 1 && ct == 4) { nw = 4; rtw = 2; }      // same KHM, same LDS layout
#endif
#define V(NWV, R, C)                                                                   \
    if (nw == NWV && rtw == R && ct == C) {                                            \
        if (cx.imat_surf)                                                              \
            return zo ? launch_variant<NWV, R, C, true, true>(cx, bn, lds, st, hip_err)         \
                      : launch_variant<NWV, R, C, false, true>(cx, bn, lds, st, hip_err);       \
        return zo ? launch_variant<NWV, R, C, true, false>(cx, bn, lds, st, hip_err)            \
                  : launch_variant<NWV, R, C, false, false>(cx, bn, lds, st, hip_err);          \
    }
    V(4, 1, 2) V(4, 2, 2) V(8, 2, 2)
    V(4, 1, 4) V(8, 1, 4)
#ifdef SOS_WIDE_REGS
    V(4, 2, 4)
#endif
#undef V
    return SOSGPU_E_UNSUPPORTED;
}
