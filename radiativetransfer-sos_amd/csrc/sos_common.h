// csrc/sos_common.h -- shared device/host structures of libsosgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// Row ordering of the 6N-long Stokes/direction state vector used by every kernel:
//   row r = c*2N + d,  c = 0:I 1:Q 2:U,  d < N: up-going direction k = d+1 (mu_k > 0)
//                                        d >= N: down-going direction k = -(d-N+1)
// The reference keeps three fields X(0:NT,-N:N) (SOS_OS.F:447-449); this is the same data with the
// unused jj = 0 column dropped.
//
// Parity decomposition of the source operator (SOS_FSOURCE_ORDREIG, SOS_OS.F:2894-2915).  With
// X^A = X(+mu) + g_c X(-mu), X^B = X(+mu) - g_c X(-mu), g = (+1,+1,-1) for (I,Q,U), the 6N x 6N operator
// splits into two independent 3N x 3N systems (A couples I^e,Q^e,U^o; B couples I^o,Q^o,U^e):
//   E^A = M^A X^A, E^B = M^B X^B,   S(+mu) = E^A + E^B,  S(-mu) = g_c (E^A - E^B)
// which halves the matrix-core work and the operator stream.  Half-system positions are listed in SosDev::rowmap:
// component-major over the directions with a non-zero Gauss weight, then the zero-weight directions (solar / user
// angles), whose operator columns vanish -- the contraction runs over the first 3 Nw positions only.
//
// Packed operator layout (A operand of v_mfma_f64_16x16x4_f64, one f64 per lane), per order s and system:
//   mp[(((s*2 + sys)*RTPH + rt)*KS2H + m)*128 + lane*2 + e] = M^sys[rt*16 + (lane&15)][8m + 2*(lane>>4) + e]
// i.e. a wave reads 1 KiB contiguous per pair of k-steps (global_load_dwordx4 per lane) and MFMA k-step
// (m,e) contracts the K indices {8m + 2q + e : q = 0..3}; the B operand is read from LDS with the same
// K permutation (one ds_read_b128 per lane per pair of k-steps).

struct SosDev {                 // per-wavelength device context, passed by value to the kernels
    int n, w, r6;               // N, 2N+1, 6N
    int kp;                     // order-1 vector stride (6N padded to 8)
    int kh, ks2h, rtph;         // half system: rows 3N padded to 8; k-pairs of the contraction ceil(3 Nw / 8); 16-row tiles ceil(kh/16)
    int nwgt;                   // Nw: directions with a non-zero quadrature weight
    int prow;                   // first of four PADDING rows (>= 3N, same 16-row tile as row 3N-1, multiple of 4) that carry the
                                // rank-4 projection V^T of the molecular operator in the packed A operand, or -1
    const int32_t *rowmap;      // [kh] half-system position -> c*N + (k-1) (weighted directions first), -1 = padding
    int os_nb, smax;            // OS_NB, iborm_max
    int n0, imat_surf, ifresnel, igmax, ipolar;
    double mus;                 // cos(solar zenith) = mu[n0-1]; the reference's TAB = -mus
    double ro;
    double beta2, gamma2, alpha2;   // molecular coefficients (SOS_OS.F:678-684), polarisation cut applied
    double f11sun, f12sun;      // Fresnel matrix at the solar incidence (SOS_OS.F:1757-1773, J=0)
    double thr_cv, thr_sum, thr_val, thr_sf;   // SOS.h:389-400
    const double *mu, *ga;      // [N]
    const double *coef;         // [4][os_nb+1] alpha,beta,gamma,zeta (polarisation cut applied)
    const double *fres;         // [3][N] F11,F12,F33 at mu_k (SOS_OS.F:1753-1780)
    double *prt;                // [smax+1][3][os_nb+1][W]  P,R,T generalised spherical functions
    double *mp_aer;             // [smax+1][2 systems][rtph*ks2h*128] parity-form aerosol operators (packed A fragments)
    double *mp_vt;              // [3][ks2h*128]  molecular operator, projection factor V^T (one 16-row tile, rows 0..3 used)
    double *mp_uf;              // [3][rtph*64]   molecular operator, expansion factor U (K = 4)
    double *sv;                 // [smax+1][4][kp]: order-1 vectors aer, ray, fresnel-aer, fresnel-ray
    // BRDF/BPDF surface (IMAT_SURF = 1), built by sosgpu_set_surface_matrices (api.hip k_pack_ground), or null:
    const double *mp_gnd;       // [smax+1][rtph*ks2h*128] ground-reflection operator G_s in the packed A-fragment layout of
                                // mp_aer (one system): G[(c,k)][(b,j)] = (2/mu_k) w_j R_cb(j -> k) (+ 2 rho w_j mu_j for
                                // c = b = 0, s = 0: the Lambertian part), rows / columns in half-system order
    const double *rdir;         // [smax+1][3][N] R_c1(N0, k): reflection of the direct solar beam (SOS_OS.F:984-990)
};

struct SosBins {
    int nb, lp;
    const int32_t *nt, *iborm, *jout;
    const double *prof, *zz;
    double *rec, *flux;
    int32_t *norders, *iglast;
    // streamed-field variant (NT too large for LDS, sos_stream.hip): per-bin scratch of scr_stride doubles
    double *scratch;
    size_t scr_stride;
    int lpb;
    int s_begin, s_end;          // streamed variant: Fourier orders s_begin <= s < s_end of this launch (order-synchronous
                                 // launches keep the source operator of the order in L2 for every workgroup of an XCD)
    unsigned long long *phase;   // diagnostic builds only (SOS_PROFILE_PHASES): [nb][8] cycle sums per phase
    // persistent, order-scheduled form of the streamed variant (sos_stream.hip, PERSIST), or null: queue[16 q] = task counter
    // of queue q (one per XCD), queue[16 (8 + q)] = finished bins of queue q; qflag[nb] = Fourier orders completed per bin.
    // All zero before the launch.
    int *queue, *qflag;
    int q_tail;                  // a queue with at most this many unfinished bins hands out whole bins (all remaining orders);
                                 // < 0: the launcher's default (twice the workgroups per queue)
    // order-parallel form of the streamed variant for few bins (sos_stream.hip): spec_k = -K set-up launch, K > 0 order tasks
    // (grid nb K: workgroup (b, j) runs order s_begin + j), 0 otherwise; spec_i3[nb][smax+1][threads] I3 terms of every order
    int spec_k;
    double *spec_i3;
    // multi-wavelength launches (sos_os_multi.hip / sos_stream_multi.hip): bin b runs with the context ctxs[ctx_of_bin[b]]
    // of a device-resident context table instead of the kernel argument; null otherwise
    const SosDev *ctxs;
    const int32_t *ctx_of_bin;
    const int32_t *order;        // multi-wavelength launches: workgroup i solves bin order[i] (costliest bins first: workgroups
                                 // are dispatched in index order and the bins stay grouped by wavelength for the aggregate), or null
};

// The kernels read the wavelength context through `cx`: the by-value kernel argument, or -- SOS_MULTI builds -- the bin's
// entry of the context table, addressed in the constant address space so that every field stays a scalar load that the
// compiler may re-issue instead of keeping it in registers (exactly what it does with kernel arguments).
typedef const __attribute__((address_space(4))) SosDev SosDevK;
typedef const __attribute__((address_space(4))) SosBins SosBinsK;
#ifdef SOS_MULTI
#define SOS_BIN_INDEX(bn) ((bn).order ? __builtin_amdgcn_readfirstlane((bn).order[blockIdx.x]) : (int)blockIdx.x)
#define SOS_BIND_CTX(cx, arg, bn)                                                                                   \
    const SosDevK &cx = *(const SosDevK *)(unsigned long long)((bn).ctxs + __builtin_amdgcn_readfirstlane((bn).ctx_of_bin[SOS_BIN_INDEX(bn)]))
#else
#define SOS_BIN_INDEX(bn) ((int)blockIdx.x)
#define SOS_BIND_CTX(cx, arg, bn) const SosDev &cx = arg
#endif
// offset of the second kernel argument (SosBins) behind the first (SosDev) in the kernarg segment: both 8-byte aligned
#define SOS_KERNARG_BINS_OFFSET ((sizeof(SosDev) + 7) & ~(size_t)7)

static inline int sos_round_up(int a, int b) { return (a + b - 1) / b * b; }
