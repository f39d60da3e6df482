// csrc/noyaux.hip -- phase-matrix Fourier kernels for every order s at once (gfx950).
//
// Replaces SOS_NOYAUX (reference src/SOS_OS.F:1857-2158), which the reference re-runs for every CKD
// bin although it depends only on the angles and the expansion coefficients.  Here it runs once per
// wavelength and writes, for each Fourier order s:
//   * the 6N x 6N source operator of SOS_FSOURCE_ORDREIG (SOS_OS.F:2894-2915) -- aerosol part, plus the
//     molecular (Rayleigh) part for s <= 2 -- packed in MFMA A-fragment order (sos_common.h);
//   * the order-1 source vectors of SOS_FSOURCE_ORDRE1 (SOS_OS.F:2520-2561) and of
//     SOS_FSOURCE_DIFF_FRESNEL1 (SOS_OS.F:3224-3292).
//
// Kernel 1 (k_gsf): generalised spherical functions P^s_l, R^s_l, T^s_l(mu_j) by the reference's
// recurrence, one thread per direction, sequential in l (SOS_OS.F:1968-2100).
// Kernel 2 (k_pack): one thread per packed operator element; the sum over l runs in the reference's
// order (l ascending) with contraction disabled so the kernels agree with the Fortran to the last bits.
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------
// prt[((s*3 + q)*(B+1) + l)*W + (jj+N)]
// ---------------------------------------------------------------------------------------------
__global__ void k_gsf(SosDev cx)
{
    const int s = blockIdx.x;
    const int j = threadIdx.x;          // 0..N ; 0 = solar beam slot, RMU(0) = -mus (SOS_OS.F:706-715)
    const int N = cx.n, W = cx.w, B = cx.os_nb;
    if (j > N) return;
    double *P = cx.prt + ((size_t)(s * 3 + 0) * (B + 1)) * W + N;
    double *R = cx.prt + ((size_t)(s * 3 + 1) * (B + 1)) * W + N;
    double *T = cx.prt + ((size_t)(s * 3 + 2) * (B + 1)) * W + N;
    const double c = (j == 0) ? -cx.mus : cx.mu[j - 1];
    // zero everything below the starting order (the reference never reads those entries)
    for (int l = 0; l <= B; l++) {
        P[(size_t)l * W + j] = 0.; R[(size_t)l * W + j] = 0.; T[(size_t)l * W + j] = 0.;
        if (j) { P[(size_t)l * W - j] = 0.; R[(size_t)l * W - j] = 0.; T[(size_t)l * W - j] = 0.; }
    }
    double pl, plm, rl, rlm, tl, tlm;   // values at l and l-1 for +j
    int lstart;
    // sg: parity factor IG of the mirror relation at the current l (SOS_OS.F:2064-2099)
    if (s == 0) {                        // SOS_OS.F:1970-1991
        const double x26 = 2. * sqrt(6.0);
        const double p2 = (3. * c * c - 1.) * 0.5;
        const double r2 = 3. * (1. - c * c) / x26;
        P[0 * W + j] = 1.; P[1 * W + j] = c; P[2 * W + j] = p2; R[2 * W + j] = r2;
        if (j) { P[0 * W - j] = 1.; P[1 * W - j] = -c; P[2 * W - j] = p2; R[2 * W - j] = r2; }
        pl = p2; plm = c; rl = r2; rlm = 0.; tl = 0.; tlm = 0.;
        lstart = 2;
    } else if (s == 1) {                 // SOS_OS.F:1999-2021
        const double x = 1. - c * c;
        const double p1 = sqrt(x * 0.5);
        const double p2 = c * p1 * sqrt(3.0);
        const double r2 = -c * sqrt(x) * 0.5;
        const double t2 = -sqrt(x) * 0.5;
        P[1 * W + j] = p1; P[2 * W + j] = p2; R[2 * W + j] = r2; T[2 * W + j] = t2;
        if (j) { P[1 * W - j] = p1; P[2 * W - j] = -p2; R[2 * W - j] = -r2; T[2 * W - j] = t2; }
        pl = p2; plm = p1; rl = r2; rlm = 0.; tl = t2; tlm = 0.;
        lstart = 2;
    } else {                             // SOS_OS.F:2027-2052
        double a = 1.;
        for (int i = 1; i <= s; i++) { double x = i; a = a * sqrt((i + s) / x) * 0.5; }
        const double b = a * sqrt(s / (s + 1.0)) * sqrt((s - 1.0) / (s + 2.));
        const double xx = 1. - c * c, yy = s * 0.5 - 1.;
        const double ps = a * pow(xx, s * 0.5);
        const double rs = b * (1. + c * c) * pow(xx, yy);
        const double ts = 2. * b * c * pow(xx, yy);
        if (s <= B) {
            P[(size_t)s * W + j] = ps; R[(size_t)s * W + j] = rs; T[(size_t)s * W + j] = ts;
            if (j) { P[(size_t)s * W - j] = ps; R[(size_t)s * W - j] = rs; T[(size_t)s * W - j] = -ts; }
        }
        pl = ps; plm = 0.; rl = rs; rlm = 0.; tl = ts; tlm = 0.;
        lstart = s;
    }
    int sg = (s == 1) ? 1 : -1;
    for (int l = lstart; l <= B - 1; l++) {   // SOS_OS.F:2067-2100
        const double a = (2 * l + 1.) / sqrt((l + s + 1.0) * (l - s + 1.));
        const double b = sqrt((double)((l + s) * (l - s))) / (2. * l + 1.);
        const double d = (l + 1.) * (2 * l + 1.) / sqrt((l + 3.0) * (l - 1.) * (l + s + 1.) * (l - s + 1.));
        const double e = sqrt((l + 2.0) * (l - 2.) * (l + s) * (l - s)) / (l * (2. * l + 1.));
        // F = 2.*IS/(L*(L+1.)) is evaluated in REAL*4 by the reference (SOS_OS.F:2079)
        const double f = (double)((2.f * (float)s) / ((float)l * ((float)l + 1.f)));
        const double pn = a * (c * pl - b * plm);
        const double rn = d * (c * rl - f * tl - e * rlm);
        const double tn = d * (c * tl - f * rl - e * tlm);
        const size_t o = (size_t)(l + 1) * W;
        P[o + j] = pn; R[o + j] = rn; T[o + j] = tn;
        if (j) { P[o - j] = sg * pn; R[o - j] = sg * rn; T[o - j] = -sg * tn; }
        plm = pl; pl = pn; rlm = rl; rl = rn; tlm = tl; tl = tn;
        sg = -sg;
    }
}

// One element of one of the six kernels of SOS_NOYAUX (SOS_OS.F:2134-2143) for order s:
//   X: 0 BP, 1 GR, 2 GT, 3 ARR, 4 ART, 5 ATT;  a, b in -N..N (0 = solar slot)
// coefficient set: aerosol (alpha..zeta arrays, l = s..B) or molecular (l = 0 and 2 only:
// beta0/beta2, gamma2, alpha2, zeta = 0 -- SOS_OS.F:2859-2876).
template <bool RAY>
__device__ inline double ktab(const SosDev &cx, int s, int X, int a, int b)
{
    const int W = cx.w, N = cx.n, B = cx.os_nb;
    const double *P = cx.prt + ((size_t)(s * 3 + 0) * (B + 1)) * W + N;
    const double *R = cx.prt + ((size_t)(s * 3 + 1) * (B + 1)) * W + N;
    const double *T = cx.prt + ((size_t)(s * 3 + 2) * (B + 1)) * W + N;
    const double *AL = cx.coef, *BE = cx.coef + (B + 1), *GA = cx.coef + 2 * (B + 1), *ZE = cx.coef + 3 * (B + 1);
    double sum = 0.;
    if (RAY) {
        if (s > 2) return 0.;
        // l = 2 term (and beta0 for BP at s = 0; P^0_0 = 1)
        const double pa = P[2 * W + a], pb = P[2 * W + b], ra = R[2 * W + a], rb = R[2 * W + b];
        const double ta = T[2 * W + a], tb = T[2 * W + b];
        switch (X) {
        case 0: return ((s == 0) ? 1. : 0.) + cx.beta2 * pa * pb;
        case 1: return cx.gamma2 * pa * rb;
        case 2: return cx.gamma2 * pa * tb;
        case 3: return cx.alpha2 * ra * rb;
        case 4: return cx.alpha2 * ta * rb;
        default: return cx.alpha2 * ta * tb;
        }
    }
    for (int l = s; l <= B; l++) {
        const size_t o = (size_t)l * W;
        switch (X) {
        case 0: sum = sum + BE[l] * P[o + a] * P[o + b]; break;
        case 1: sum = sum + GA[l] * P[o + a] * R[o + b]; break;
        case 2: sum = sum + GA[l] * P[o + a] * T[o + b]; break;
        case 3: { double r1 = T[o + a] * T[o + b], r2 = R[o + a] * R[o + b]; sum = sum + ZE[l] * r1 + AL[l] * r2; } break;
        case 4: sum = sum + AL[l] * R[o + b] * T[o + a] + ZE[l] * R[o + a] * T[o + b]; break;
        default: { double r1 = T[o + a] * T[o + b], r2 = R[o + a] * R[o + b]; sum = sum + AL[l] * r1 + ZE[l] * r2; } break;
        }
    }
    return sum;
}

// Half-system operator element (parity decomposition, sos_common.h).  sg = +1 for system A, -1 for B;
// plus(X,a,b) = X(a,b) + sg X(a,-b), minus(X,a,b) = X(a,b) - sg X(a,-b); rows/cols kk = c*N + (k-1):
//   I<-I  plus(BP,j,k)    I<-Q  plus(GR,k,j)     I<-U  -minus(GT,k,j)
//   Q<-I  plus(GR,j,k)    Q<-Q  plus(ARR,j,k)    Q<-U  -plus(ART,j,k)
//   U<-I  -minus(GT,j,k)  U<-Q  -plus(ART,k,j)   U<-U  minus(ATT,j,k)
// times w_j/4 (the reference's Gauss weight and final 1/2, and the 1/2 of the recombination).
template <bool RAY>
__device__ inline double half_element(const SosDev &cx, int s, int sys, int row, int col)
{
    const int N = cx.n;
    const int ro = cx.rowmap[row], cl = cx.rowmap[col];         // half-system positions -> (component, direction)
    const int co = ro / N, k = ro % N + 1, ci = cl / N, j = cl % N + 1;
    const double sg = sys ? -1. : 1.;
    int X, a, b;
    bool mns = false, neg = false;
    switch (co * 3 + ci) {
    case 0: X = 0; a = j; b = k; break;
    case 1: X = 1; a = k; b = j; break;
    case 2: X = 2; a = k; b = j; mns = true; neg = true; break;
    case 3: X = 1; a = j; b = k; break;
    case 4: X = 3; a = j; b = k; break;
    case 5: X = 4; a = j; b = k; neg = true; break;
    case 6: X = 2; a = j; b = k; mns = true; neg = true; break;
    case 7: X = 4; a = k; b = j; neg = true; break;
    default: X = 5; a = j; b = k; mns = true; break;
    }
    const double same = ktab<RAY>(cx, s, X, a, b), opp = ktab<RAY>(cx, s, X, a, -b);
    double v = mns ? (same - sg * opp) : (same + sg * opp);
    if (neg) v = -v;
    return 0.25 * cx.ga[j - 1] * v;
}

// Element (row 0..3, half-system position col) of the projection factor V^T of the molecular operator (see k_pack_ray)
__device__ inline double ray_vt_element(const SosDev &cx, int s, int row, int col)
{
    const int N = cx.n, W = cx.w, B = cx.os_nb;
    const double *P = cx.prt + ((size_t)(s * 3 + 0) * (B + 1) + 2) * W + N;   // l = 2
    const double *R = cx.prt + ((size_t)(s * 3 + 1) * (B + 1) + 2) * W + N;
    const double *T = cx.prt + ((size_t)(s * 3 + 2) * (B + 1) + 2) * W + N;
    const double b0 = (s == 0) ? 1. : 0., b2 = cx.beta2, g2 = cx.gamma2, a2 = cx.alpha2;
    const int cl = cx.rowmap[col];
    const int ci = cl / N, j = cl % N + 1;
    const double hw = 0.5 * cx.ga[j - 1];
    const double f = (ci == 0) ? P[j] : (ci == 1 ? R[j] : T[j]);
    const double c0[3] = {b2, g2, -g2}, c1[3] = {g2, a2, -a2}, c2[3] = {-g2, -a2, a2};
    if (row == 0) return hw * c0[ci] * f;
    if (row == 1) return hw * c1[ci] * f;
    if (row == 2) return hw * c2[ci] * f;
    return (ci == 0) ? hw * b0 : 0.;
}

__global__ void k_pack(SosDev cx)
{
    const int s = blockIdx.y;
    const size_t per = (size_t)cx.rtph * cx.ks2h * 128;     // one system
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * per) return;
    const int sys = e >= per;
    const size_t q = e - (size_t)sys * per;
    const int e2 = q & 1, lane = (q >> 1) & 63;
    const int m = (int)((q >> 7) % cx.ks2h), rt = (int)((q >> 7) / cx.ks2h);
    const int row = rt * 16 + (lane & 15);
    const int col = 8 * m + 2 * (lane >> 4) + e2;
    const int H = 3 * cx.n;
    const bool in = row < H && col < H;
    double v = in ? half_element<false>(cx, s, sys, row, col) : 0.;
    // the four projection rows of the rank-4 molecular operator ride in padding rows of the half system they act on
    // (A for even s, B for odd s): the dense contraction then delivers V^T X for free
    if (cx.prow >= 0 && s <= 2 && sys == (s & 1) && row >= cx.prow && row < cx.prow + 4 && col < H)
        v = ray_vt_element(cx, s, row - cx.prow, col);
    cx.mp_aer[(size_t)s * 2 * per + e] = v;
}

// Molecular (Rayleigh) part of the source operator for s <= 2 (SOS_OS.F:2859-2876).  Its kernels are single
// Legendre terms (l = 2, plus beta0 at s = 0), so in the parity basis the operator is EXACTLY rank <= 4 and acts on
// one half system only (A for even s, B for odd s; the other half is identically zero):
//     M_ray = U V^T,   V^T [4 x 3N] (rows: I-, Q-, U-row projections, beta0 sum),   U [3N x 4]
//   V^T[0] = w/2 ( b2 P,  g2 R, -g2 T)   -> times P(k) on the I rows
//   V^T[1] = w/2 ( g2 P,  a2 R, -a2 T)   -> times R(k) on the Q rows
//   V^T[2] = w/2 (-g2 P, -a2 R,  a2 T)   -> times T(k) on the U rows
//   V^T[3] = w/2 ( b0, 0, 0)             -> times 1 on the I rows (s = 0)
// with P,R,T = P^s_2, R^s_2, T^s_2 at the Gauss nodes.  Packed as MFMA A operands:
//   vt[(m*64 + lane)*2 + e] = V^T[lane&15][8m + 2(lane>>4) + e]   (projection, K = 3N)
//   uf[rt*64 + lane]        = U[rt*16 + (lane&15)][lane>>4]       (expansion, K = 4)
__global__ void k_pack_ray(SosDev cx)
{
    const int s = blockIdx.x;                 // 0..2
    const int N = cx.n, W = cx.w, B = cx.os_nb;
    const double *P = cx.prt + ((size_t)(s * 3 + 0) * (B + 1) + 2) * W + N;   // l = 2
    const double *R = cx.prt + ((size_t)(s * 3 + 1) * (B + 1) + 2) * W + N;
    const double *T = cx.prt + ((size_t)(s * 3 + 2) * (B + 1) + 2) * W + N;
    double *vt = cx.mp_vt + (size_t)s * cx.ks2h * 128;
    double *uf = cx.mp_uf + (size_t)s * cx.rtph * 64;
    for (int e = threadIdx.x; e < cx.ks2h * 128; e += blockDim.x) {
        const int e2 = e & 1, lane = (e >> 1) & 63, m = e >> 7;
        const int row = lane & 15, col = 8 * m + 2 * (lane >> 4) + e2;
        const double v = (row < 4 && col < 3 * N) ? ray_vt_element(cx, s, row, col) : 0.;
        vt[e] = v;
    }
    for (int e = threadIdx.x; e < cx.rtph * 64; e += blockDim.x) {
        const int lane = e & 63, rt = e >> 6;
        const int row = rt * 16 + (lane & 15), k4 = lane >> 4;
        double v = 0.;
        if (row < 3 * N) {
            const int ro = cx.rowmap[row];
            const int co = ro / N, k = ro % N + 1;
            if (co == 0) v = (k4 == 0) ? P[k] : (k4 == 3 ? ((s == 0) ? 1. : 0.) : 0.);
            else if (co == 1) v = (k4 == 1) ? R[k] : 0.;
            else v = (k4 == 2) ? T[k] : 0.;
        }
        uf[e] = v;
    }
}

// Order-1 source vectors, one per state row r = (c, +-k):
//   sv[s][0][r]  aerosol part of SOS_FSOURCE_ORDRE1:  I: BP(0,J), Q: GR(0,J), U: -GT(0,J)   (SOS_OS.F:2557-2559)
//   sv[s][1][r]  molecular part (s <= 2)
//   sv[s][2][r]  aerosol part of SOS_FSOURCE_DIFF_FRESNEL1 for the field of direction J = +-k, which
//                uses the mirrored direction D = -J (SOS_OS.F:3280-3289):
//                I: F11sun BP(0,D) + F12sun GR(D,0), Q: F11sun GR(0,D) + F12sun ARR(0,D),
//                U: F11sun GT(0,D) + F12sun ART(D,0)
//   sv[s][3][r]  molecular part of the same (s <= 2)
__global__ void k_sv(SosDev cx)
{
    const int s = blockIdx.x;
    const int r = threadIdx.x;
    if (r >= cx.kp) return;
    double *o = cx.sv + (size_t)s * 4 * cx.kp;
    double v0 = 0., v1 = 0., v2 = 0., v3 = 0.;
    if (r < cx.r6) {
        const int N = cx.n;
        const int c = r / (2 * N), d = r % (2 * N);
        const int J = d < N ? d + 1 : -(d - N + 1);
        const int D = -J;
        const double f11 = cx.f11sun, f12 = cx.f12sun;
        if (c == 0) {
            v0 = ktab<false>(cx, s, 0, 0, J); v1 = ktab<true>(cx, s, 0, 0, J);
            v2 = f11 * ktab<false>(cx, s, 0, 0, D) + f12 * ktab<false>(cx, s, 1, D, 0);
            v3 = f11 * ktab<true>(cx, s, 0, 0, D) + f12 * ktab<true>(cx, s, 1, D, 0);
        } else if (c == 1) {
            v0 = ktab<false>(cx, s, 1, 0, J); v1 = ktab<true>(cx, s, 1, 0, J);
            v2 = f11 * ktab<false>(cx, s, 1, 0, D) + f12 * ktab<false>(cx, s, 3, 0, D);
            v3 = f11 * ktab<true>(cx, s, 1, 0, D) + f12 * ktab<true>(cx, s, 3, 0, D);
        } else {
            v0 = -ktab<false>(cx, s, 2, 0, J); v1 = -ktab<true>(cx, s, 2, 0, J);
            v2 = f11 * ktab<false>(cx, s, 2, 0, D) + f12 * ktab<false>(cx, s, 4, D, 0);
            v3 = f11 * ktab<true>(cx, s, 2, 0, D) + f12 * ktab<true>(cx, s, 4, D, 0);
        }
    }
    o[0 * cx.kp + r] = v0; o[1 * cx.kp + r] = v1; o[2 * cx.kp + r] = v2; o[3 * cx.kp + r] = v3;
}

// Parity accessor: the six (W x W) kernels of one order laid out as the reference does.
__global__ void k_noyaux_fetch(SosDev cx, int s, double *out)
{
    const int W = cx.w, N = cx.n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < W * W) {
        const int j = idx / W - N, k = idx % W - N;
        for (int X = 0; X < 6; X++) out[(size_t)X * W * W + idx] = ktab<false>(cx, s, X, j, k);
    }
    if (idx < W) {
        const int B = cx.os_nb;
        for (int q = 0; q < 3; q++)
            out[(size_t)6 * W * W + q * W + idx] = (B >= 2) ? cx.prt[((size_t)(s * 3 + q) * (B + 1) + 2) * W + idx] : 0.;
    }
}

void launch_noyaux(const SosDev &cx, hipStream_t st)
{
    const int S = cx.smax + 1;
    k_gsf<<<S, 128, 0, st>>>(cx);
    const size_t per = (size_t)2 * cx.rtph * cx.ks2h * 128;
    dim3 g((unsigned)((per + 255) / 256), S);
    k_pack<<<g, 256, 0, st>>>(cx);
    k_pack_ray<<<(cx.smax < 2 ? cx.smax + 1 : 3), 256, 0, st>>>(cx);
    k_sv<<<S, sos_round_up(cx.kp, 64), 0, st>>>(cx);
}

void launch_noyaux_fetch(const SosDev &cx, int s, double *d_out, hipStream_t st)
{
    const int n = cx.w * cx.w;
    k_noyaux_fetch<<<(n + 255) / 256, 256, 0, st>>>(cx, s, d_out);
}
