// csrc/profile.hip -- per-bin atmospheric profile discretisation on the device (gfx950).
//
// Replaces, for every CKD bin of a wavelength at once, the step immediately before the solver:
//   SOS_PROFILE (IPROFIL = 1)  src/SOS_PROFIL.F:224-1170   gas step :509-795 (level placement by optical-depth steps,
//                                                            forced levels of the no-gas profile, limit layer for
//                                                            strong absorption), PROFIL file formats :1084,1150
//   SOS_DISC                   src/SOS_PROFIL.F:1210-1332  bisection on altitude
//   SOS (wrapper)              src/SOS.F:511-550           read-back of the PROFIL file, truncation rescale, IBORM
//                              src/SOS.F:567-589           TAUOUT / TTOT_TRONC; output level of SOS_OS.F:1514-1520
// The reference writes one PROFIL text file per bin and reads it back; here the profile goes straight into the
// device layout of sosgpu_os_solve and the text round trip (F10.5 / E15.8) is applied in registers, exactly.
//
// Work split: the no-gas profile (SOS_PROFIL.F:349-489) depends only on (TR,HR,TA,HA) -- identical for all bins of a
// wavelength -- and is computed once, by one wavefront (k_profile_nogas, queued in front of k_profile); the gas step depends
// on the bin's absorption profile and runs one wavefront per bin (the level loop is a serial recurrence: each level starts
// from the optical depth the previous bisection actually reached).  Latency/transcendental-bound: ~NT * 25 bisection steps * 2 exp per bin, no
// HBM traffic to speak of (50 doubles in, 4 * (NT+1) doubles out per bin).
#include <algorithm>
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace {

__device__ const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                   1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// round(x * p) to the nearest integer (ties to even) where p is an exact power of ten: the product is formed exactly as
// hi + lo (one FMA), so the decision never sees the rounding error of the multiplication
__device__ __forceinline__ double round_scaled(double x, double p)
{
    const double hi = x * p;
    const double lo = __builtin_fma(x, p, -hi);
    double r = rint(hi);
    const double d = (hi - r) + lo;                  // exact: |hi - r| <= 0.5, lo tiny
    if (d > 0.5 || (d == 0.5 && fmod(r, 2.0) != 0.0)) r += 1.0;
    else if (d < -0.5 || (d == -0.5 && fmod(r, 2.0) != 0.0)) r -= 1.0;
    return r;
}

// value read back from a Fortran E15.8 field: 8 significant decimal digits, correctly rounded both ways
// (decimal -> binary is r / 10^m with r < 2^27 and 10^m exact: one correctly rounded division)
__device__ double rt_e15_8(double v)
{
    if (v == 0.0 || !isfinite(v)) return v;
    const double av = fabs(v);
    int e = (int)floor(log10(av));
    // fix the estimate with exact comparisons where the powers are exact
    if (e >= 0 && e < 22) { if (av >= P10[e + 1]) e++; else if (av < P10[e]) e--; }
    else if (e < 0 && e >= -22) { if (av * P10[-e] < 1.0) e--; else if (e < -1 && av * P10[-e - 1] >= 1.0) e++; }
    const int m = 7 - e;                             // scale so that the integer has 8 digits
    double r;
    if (m >= 0 && m <= 22) {
        r = round_scaled(av, P10[m]);
        if (r >= 1e8) return copysign(r / P10[m], v);            // carried into the next decade: same value
        if (r < 1e7) {                                          // estimate one too high (only at exact powers of ten)
            r = round_scaled(av, P10[m] * 10.0);
            return copysign(r / (P10[m] * 10.0), v);
        }
        return copysign(r / P10[m], v);
    }
    if (m < 0) {                                     // |v| >= 1e8: divide
        r = rint(av / P10[-m]);
        return copysign(r * P10[-m], v);
    }
    // |v| < 1e-15: 10^m is not exact any more; these are fractions of 1e-15 of a layer, far below the parity bar
    const double p = P10[22] * pow(10.0, (double)(m - 22));
    r = rint(av * p);
    return copysign(r / p, v);
}

__device__ __forceinline__ double rt_f10_5(double v) { return copysign(round_scaled(fabs(v), 1e5) / 1e5, v); }

struct GasProf {
    const double *alt, *tab;     // altitude grid (descending) and cumulative gas optical depth of the bin, in LDS
    int n;
    // index J of the absorption-profile segment holding altitude z: first J >= 2 with z >= ALTABS(J), i.e.
    // ALTABS(J-1) > z >= ALTABS(J) (SOS_PROFIL.F:694-697).  The reference scans J upwards; the grid is monotonic, so a
    // bisection returns the same J in 6 steps instead of up to 50 dependent loads.
    __device__ __forceinline__ int seg(double z) const
    {
        int lo = 2, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (z < alt[mid - 1]) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
};

// SOS_DISC (SOS_PROFIL.F:1260-1325)
__device__ double disc(double dt, double ta, double ha, double tr, double hr, const GasProf &g, double tim1,
                       double zmax_init, double tg_zlim, double zlim)
{
    const double ti = tim1 + dt;
    double zmax = zmax_init, zmin = zlim, zmoy;
    for (int guard = 0; guard < 4096; guard++) {
        double tg;
        zmoy = (zmax + zmin) / 2.;
        if (tg_zlim > 0.0) {
            const int j = g.seg(zmoy);
            double zz;
            if (zmoy > g.alt[0]) zz = 0;
            else zz = (zmoy - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
            tg = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
        } else tg = 0.0;
        const double tzmoy = ta * exp(-zmoy / ha) + tr * exp(-zmoy / hr) + tg;
        const double xd = fabs(ti - tzmoy);
        if (xd < (double).000001f) break;
        if (zmoy == 0.0) break;
        if ((ti - tzmoy) < 0.0) zmin = zmoy; else zmax = zmoy;
    }
    return zmoy;
}

// SOS_DISC by the 64 lanes of a wavefront: the same bisection, six of its steps per evaluation.  The bisection visits a path
// in the binary tree of brackets below [zmin, zmax]; lane n = 1..63 takes heap node n of the six-level subtree under the
// current bracket, walks to it with the reference's own midpoint statement (so its ZMOY is bit for bit the one the serial loop
// would form on that path), and evaluates the optical depth there -- the 2 exponentials and the gas-profile lookup that make a
// serial step ~1000 cycles.  The comparison and stop results of all nodes are two wave masks; the path through them is resolved
// with scalar bit tests.  25-30 serial evaluations per level become 5 rounds.
__device__ __forceinline__ double lane_read(double v, int lane)
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

__device__ double disc_wave(double dt, double ta, double ha, double tr, double hr, const GasProf &g, double tim1,
                            double zmax_init, double tg_zlim, double zlim)
{
    const double ti = tim1 + dt;
    const int lane = threadIdx.x & 63;
    const int node = lane ? lane : 1;                        // lane 0 shadows the root (its results are not used)
    const int depth = 31 - __builtin_clz(node);              // 0..5
    double zmax = zmax_init, zmin = zlim, zres = 0.;
    for (int round = 0; round < 683; round++) {              // 683 x 6 >= the 4096 evaluations of the serial guard
        double lo = zmin, hi = zmax, zmoy = (hi + lo) / 2.;
        for (int k = depth - 1; k >= 0; --k) {               // the path to this lane's node: bit set = "ZMIN = ZMOY" was taken
            if ((node >> k) & 1) lo = zmoy; else hi = zmoy;
            zmoy = (hi + lo) / 2.;
        }
        double tg;
        if (tg_zlim > 0.0) {
            const int j = g.seg(zmoy);
            double zz;
            if (zmoy > g.alt[0]) zz = 0;
            else zz = (zmoy - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
            tg = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
        } else tg = 0.0;
        const double tzmoy = ta * exp(-zmoy / ha) + tr * exp(-zmoy / hr) + tg;
        const double xd = fabs(ti - tzmoy);
        const bool stop = xd < (double).000001f || zmoy == 0.0;
        const bool up = (ti - tzmoy) < 0.0;                  // ZMIN = ZMOY
        const unsigned long long sm = __ballot(stop), um = __ballot(up);
        int n = 1;
        bool found = false;
        for (int d = 0; d < 6; d++) {
            if ((sm >> n) & 1ull) { found = true; break; }
            if (d == 5) break;
            n = 2 * n + (int)((um >> n) & 1ull);
        }
        n = __builtin_amdgcn_readfirstlane(n);
        zres = lane_read(zmoy, n);
        if (found) return zres;
        // six steps taken without a stop: the bracket below node n (a node of the last level)
        const double lo_n = lane_read(lo, n), hi_n = lane_read(hi, n);
        if ((um >> n) & 1ull) { zmin = zres; zmax = hi_n; } else { zmin = lo_n; zmax = zres; }
    }
    return zres;
}

// The first level below the top of the atmosphere (SOS_PROFIL.F:424-433 without gas, :601-624 with): Z = Z - DELTA_Z from Z0
// until the optical depth above Z reaches T_FIRST -- ~1700 steps of two exponentials each down to 35 km, the reference's serial
// scan, and half of the time this file's kernels took.  By the 64 lanes of a wavefront: a block of 64 consecutive steps at a
// time, lane l forming ITS altitude by the same l + 1 subtractions the serial loop makes (so it is bit for bit the loop's
// value) and evaluating the optical depth there; the first lane whose depth is not below T_FIRST is where the loop stops.
template <bool GAS>
__device__ double scan_first_wave(double tr, double hr, double ta, double ha, const GasProf &g, double t_first, double z0,
                                  double delta_z)
{
    const int lane = threadIdx.x & 63;
    double zbase = z0;
    for (int blk = 0; blk < 65536; blk++) {                  // (the serial loop has no guard either: TAU(Z) reaches T_FIRST or Z = -inf)
        double z = zbase;
        for (int j = 0; j <= lane; j++) z = z - delta_z;     // lane l: step blk * 64 + l + 1 of the serial loop
        double dtau;
        if (GAS) {
            double vg;
            const int j = g.seg(z);
            if (z <= g.alt[0]) {
                const double zz = (z - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
                vg = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
            } else vg = 0.;
            const double vr = tr * exp(-z / hr), va = ta * exp(-z / ha);
            dtau = vr + va + vg;
        } else dtau = tr * exp(-z / hr) + ta * exp(-z / ha);
        const unsigned long long stop = __ballot(!(dtau < t_first));
        if (stop) return lane_read(z, __builtin_ctzll(stop));
        zbase = lane_read(z, 63);
    }
    return zbase;
}

}  // namespace

// The no-gas profile of the wavelength (SOS_PROFIL.F:349-489): level altitudes by equal optical-depth steps of molecules +
// aerosols, the shares of the two in every layer.  One wavefront, all lanes in lockstep on the same values (stores of one
// instruction go to one address), SOS_DISC by the 64 lanes (disc_wave).  Rounds 1-2 ran this loop on the host inside
// sosgpu_profile -- NT x 25 bisection steps x 2 exp, 0.3-0.4 ms of every wavelength of a spectrum plus a waited-for upload;
// the level count and the two steps are closed forms of (TR, TA) and stay there (api.hip, profile_nogas_grid).
//   z, h, pca, pcm [nt + 1]: altitude, cumulative optical depth, aerosol and molecular share of the layer ending at the level
__global__ __launch_bounds__(64) void k_profile_nogas(double tr, double hr, double ta, double ha, int nt, double t_first,
                                                      double t_layer, double *__restrict__ z, double *__restrict__ h,
                                                      double *__restrict__ pca, double *__restrict__ pcm)
{
    const double TOA = 120.0, DELTA_Z = (double)0.05f;
    if (ta == 0.0) {                                             // molecules only: closed form (SOS_PROFIL.F:367-392)
        for (int i = threadIdx.x; i <= nt; i += 64) {
            const double hm = i == 0 ? 0. : (i == 1 ? t_first : (i - 1) * t_layer + t_first);
            pcm[i] = 1.; pca[i] = 0.;
            z[i] = i == 0 ? TOA : hr * log(tr / hm);
            h[i] = hm + 0.;
        }
        return;
    }
    GasProf g;
    g.n = 0; g.alt = nullptr; g.tab = nullptr;                   // (never read: TG_ZLIM = 0 below)
    double zz = scan_first_wave<false>(tr, hr, ta, ha, g, t_first, TOA, DELTA_Z);
    const double dtau = tr * exp(-zz / hr) + ta * exp(-zz / ha);
    double vr = tr * exp(-zz / hr), va = ta * exp(-zz / ha);
    double hmol_p = vr, haer_p = va;
    const double z1 = zz;
    z[0] = TOA; h[0] = 0.;
    z[1] = zz; h[1] = vr + va;
    pcm[1] = vr / dtau; pca[1] = va / dtau;
    pcm[0] = pcm[1]; pca[0] = pca[1];
    double hprev = dtau;
    for (int i = 2; i <= nt - 1; i++) {
        zz = disc_wave(t_layer, ta, ha, tr, hr, g, hprev, z1, 0.0, 0.0);
        z[i] = zz;
        vr = tr * exp(-zz / hr); va = ta * exp(-zz / ha);
        const double hm = vr, hae = va;
        hprev = vr + va;
        h[i] = hprev;
        vr = vr - hmol_p; va = va - haer_p;
        pcm[i] = vr / (vr + va); pca[i] = va / (vr + va);
        hmol_p = hm; haer_p = hae;
    }
    z[nt] = 0.; h[nt] = tr + ta;
    vr = tr - hmol_p; va = ta - haer_p;
    pcm[nt] = vr / (vr + va); pca[nt] = va / (vr + va);
}

void launch_profile_nogas(double tr, double hr, double ta, double ha, int nt, double t_first, double t_layer, double *d_ng, int ng,
                          hipStream_t st)
{
    k_profile_nogas<<<1, 64, 0, st>>>(tr, hr, ta, ha, nt, t_first, t_layer, d_ng, d_ng + ng, d_ng + 2 * ng, d_ng + 3 * ng);
}

// One thread per bin, `bpw` bins per wavefront.  The work of a bin is a long serial chain (latency-bound) and very ragged
// (NT 100...600, bisection depth varies), so a wavefront is deliberately left mostly EMPTY: with few bins per wave the
// batch spreads over all 1024 SIMDs and a wave only waits for the slowest of its few bins.
// prof[b][3][lp] <- H, XDEL, YDEL (after the rescale), zprof[b][lp], nt, iborm, jout, zz,
// scal[b][4] = {0, TTOT_TRONC, TTOT_VRAI, TAUOUT}; nt[b] = -1 flags a profile that does not fit (IER of the reference).
// WAVE = true (the form launched): ONE WAVEFRONT per bin.  All 64 lanes run the bin's level loop in lockstep on the same
// values (stores of one instruction go to one address) and share the work where the time is: SOS_DISC (disc_wave).  A single
// bin takes 6.5-9.7 ms with one lane (most of the latency of a 25-bin sos_proc call) and 2.3-2.9 ms this way; 4096 bins 24.8 -> 3.8-5.3 ms.
template <bool WAVE>
__global__ __launch_bounds__(64) void k_profile(ProfileArgs a, int bpw)
{
    if (!WAVE && (int)threadIdx.x >= bpw) return; // no barrier below: every thread only touches its own LDS slice
    const int b = WAVE ? (int)blockIdx.x : blockIdx.x * bpw + threadIdx.x;
    const int slot = WAVE ? 0 : (int)threadIdx.x;
    if (b >= a.nb) return;
    const double TCOUCHE = (double)0.005f, T_FIRST = (double)0.0002f, DELTA_Z = (double)0.05f, DZ = (double)0.001f;
    const double TAUABS_MAX = 1.5, TOA = 120.0;
    const int OS_NT = 600, OS_NT_MIN = 100;
    const double tr = a.tr, hr = a.hr, ta = a.ta, ha = a.ha;
    double *H = a.prof + (size_t)b * 3 * a.lp, *XD = H + a.lp, *YD = XD + a.lp;
    double *Z = a.zprof + (size_t)b * a.lp;
    // the bin's absorption profile is read thousands of times: keep it (and the altitude grid) in LDS
    __shared__ double s_alt[SOS_PROF_NBLEV_MAX];
    __shared__ double s_tab[WAVE ? 1 : 64][SOS_PROF_NBLEV_MAX + 1];     // +1: odd stride, lanes on different banks
    GasProf g;
    g.n = a.nblev; g.alt = s_alt; g.tab = nullptr;
    if (a.tabs) {
        for (int i = 0; i < a.nblev; i++) { s_tab[slot][i] = a.tabs[(size_t)b * a.nblev + i]; s_alt[i] = a.altabs[i]; }
        g.tab = s_tab[slot];
    }
    const double tgtot = g.tab ? g.tab[a.nblev - 1] : 0.0;
    int nt;
    bool bad = a.nt_ng < 1 || a.nt_ng > OS_NT || a.nt_ng >= a.lp;

    if (bad) {
        nt = 0;
    } else if (a.absprofil == 7 || tgtot == 0.0) {                    // SOS_PROFIL.F:492-508
        nt = a.nt_ng;
        for (int i = 0; i <= nt; i++) { Z[i] = a.z_ng[i]; H[i] = a.h_ng[i]; XD[i] = a.pca_ng[i]; YD[i] = a.pcm_ng[i]; }
    } else {
        // ---- profile with gas absorption (SOS_PROFIL.F:509-795); Hmol/Haer/Habs of the previous level in registers
        const bool strong = tgtot > TAUABS_MAX;
        double t_first, t_layer, zlim, tg_zlim, ttot_zlim;
        const double ttot = tr + ta + tgtot;
        if (strong) {
            int i = 1;
            while (g.tab[i - 1] < TAUABS_MAX) i++;
            const double alin = (g.tab[i - 1] - g.tab[i - 2]) / (g.alt[i - 1] - g.alt[i - 2]);
            const double blin = g.tab[i - 1] - alin * g.alt[i - 1];
            tg_zlim = TAUABS_MAX;
            zlim = (tg_zlim - blin) / alin;
            t_first = T_FIRST;
            ttot_zlim = ta * exp(-zlim / ha) + tr * exp(-zlim / hr) + tg_zlim;
            t_layer = (ttot_zlim - t_first) / (OS_NT - a.nt_ng - 2);
            t_layer = fmax(t_layer, TCOUCHE);
        } else {
            zlim = 0.; tg_zlim = tgtot;
            if ((ttot / OS_NT_MIN) <= T_FIRST) { t_layer = ttot / OS_NT_MIN; t_first = t_layer; }
            else if ((ttot / OS_NT_MIN) < TCOUCHE) { t_first = T_FIRST; t_layer = (ttot - t_first) / OS_NT_MIN; }
            else { t_first = T_FIRST; const int n0 = (int)((ttot - t_first) / TCOUCHE); t_layer = (ttot - t_first) / n0; }
        }
        nt = 1;
        double z = TOA, zing = a.z_ng[1];
        double hmol_p = 0., haer_p = 0., habs_p = 0., h_p = 0.;   // level nt-1
        int ing = 1;
        ttot_zlim = ta * exp(-zlim / ha) + tr * exp(-zlim / hr) + tg_zlim;
        H[0] = 0.;
        while ((ttot_zlim - h_p) > t_layer) {
            const int i = nt;
            if (i > OS_NT - 1 || i >= a.lp - 1) { bad = true; break; }
            double vr, va, vg;
            if (i == 1 && WAVE) {
                z = scan_first_wave<true>(tr, hr, ta, ha, g, t_first, z, DELTA_Z);
                ing = 1;
            } else if (i == 1) {
                double dtau = 0.;
                while (dtau < t_first) {
                    z = z - DELTA_Z;
                    const int j = g.seg(z);
                    if (z <= g.alt[0]) {
                        const double zz = (z - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
                        vg = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
                    } else vg = 0.;
                    vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
                    dtau = vr + va + vg;
                }
                ing = 1;
            } else {
                z = WAVE ? disc_wave(t_layer, ta, ha, tr, hr, g, h_p, Z[1], tg_zlim, zlim)
                         : disc(t_layer, ta, ha, tr, hr, g, h_p, Z[1], tg_zlim, zlim);
            }
            if (z <= zing) { z = zing; ing = ing + 1; zing = a.z_ng[min(ing, a.nt_ng)]; }
            else if ((z - zing) <= DZ) { ing = ing + 1; zing = a.z_ng[min(ing, a.nt_ng)]; }
            Z[i] = z;
            const int j = g.seg(z);
            if (z > g.alt[0]) vg = g.tab[j - 2];
            else {
                const double zz = (z - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
                vg = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
            }
            vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
            const double hm = vr, hae = va, hab = vg;
            h_p = va + vr + vg;
            H[i] = h_p;
            va = va - haer_p; vr = vr - hmol_p; vg = vg - habs_p;
            XD[i] = va / (va + vr + vg); YD[i] = vr / (va + vr + vg);
            hmol_p = hm; haer_p = hae; habs_p = hab;
            nt = nt + 1;
        }
        if (!bad) {
            // limit level (ground, or the altitude where the gas optical depth reaches the threshold)
            double hmol_q = hmol_p, haer_q = haer_p, habs_q = habs_p;      // level nt-1 ...
            if ((Z[nt - 1] - zlim) <= DZ) {
                // the last computed level is dropped: level nt-1 is the one before it -- recompute its parts from Z
                nt = nt - 1;
                const double zq = Z[nt - 1];
                if (nt - 1 == 0) { hmol_q = 0.; haer_q = 0.; habs_q = 0.; }
                else {
                    hmol_q = tr * exp(-zq / hr); haer_q = ta * exp(-zq / ha);
                    const int j = g.seg(zq);
                    if (zq > g.alt[0]) habs_q = g.tab[j - 2];
                    else {
                        const double zz = (zq - g.alt[j - 2]) / (g.alt[j - 1] - g.alt[j - 2]);
                        habs_q = (1 - zz) * g.tab[j - 2] + zz * g.tab[j - 1];
                    }
                }
            }
            Z[nt] = zlim;
            double vr = tr * exp(-zlim / hr), va = ta * exp(-zlim / ha), vg = tg_zlim;
            const double hm = vr, hae = va, hab = vg;
            H[nt] = vr + va + tg_zlim;
            va = va - haer_q; vr = vr - hmol_q; vg = vg - habs_q;
            XD[nt] = va / (va + vr + vg); YD[nt] = vr / (va + vr + vg);
            Z[0] = TOA; XD[0] = XD[1]; YD[0] = YD[1]; H[0] = 0.;
            if (strong) {
                nt = nt + 1;
                if (nt > OS_NT || nt >= a.lp) bad = true;
                else {
                    H[nt] = (tr + ta) + tgtot;
                    vr = tr - hm; va = ta - hae; vg = tgtot - hab;
                    XD[nt] = va / (va + vr + vg); YD[nt] = vr / (va + vr + vg);
                    Z[nt] = 0.;
                }
            }
        }
    }
    if (bad) {
        a.nt[b] = -1; a.iborm[b] = 0;
        if (a.jout) { a.jout[b] = 0; a.zz[b] = 0.; }
        for (int k = 0; k < 4; k++) a.scal[4 * b + k] = 0.;
        return;
    }
    // PROFIL file round trip (SOS_PROFIL.F:1084 format 20 -> SOS.F:515 format 70)
    // (WAVE: the independent per-level passes are dealt to the lanes, level i to lane i mod 64; the barrier of the one-wave
    //  block makes the stores of the other lanes visible to the serial passes that follow)
    const int i0 = WAVE ? (int)(threadIdx.x & 63) : 0, di = WAVE ? 64 : 1;
    if (WAVE) __syncthreads();
    for (int i = i0; i <= nt; i += di) { Z[i] = rt_f10_5(Z[i]); H[i] = rt_e15_8(H[i]); XD[i] = rt_e15_8(XD[i]); YD[i] = rt_e15_8(YD[i]); }
    if (WAVE) __syncthreads();
    const double ttot_vrai = H[nt];
    // truncation rescale (SOS.F:521-543) and IBORM (:549-550)
    bool lta = true;
    if (a.a_tronc != 0.) {
        double htr_p = H[0], h_prev = H[0];
        for (int i = 1; i <= nt; i++) {
            const double dh = H[i] - h_prev;
            h_prev = H[i];
            const double va = XD[i] * dh;
            const double vatr = va * (1 - a.piz * 0.5 * a.a_tronc);
            const double vr = YD[i] * dh;
            const double vg = (1 - XD[i] - YD[i]) * dh;
            const double htr = (vatr + vr + vg) + htr_p;
            XD[i] = vatr / (vatr + vr + vg);
            YD[i] = vr / (vatr + vr + vg);
            H[i] = htr;
            htr_p = htr;
        }
    }
    if (WAVE) __syncthreads();
    for (int i = i0; i <= nt; i += di) { XD[i] = XD[i] * a.piztr; if (XD[i] != 0.) lta = false; }
    if (WAVE) lta = !__any(!lta);
    for (int i = nt + 1 + i0; i < a.lp; i += di) { H[i] = 0.; XD[i] = 0.; YD[i] = 0.; Z[i] = 0.; }
    if (WAVE) __syncthreads();
    a.nt[b] = nt;
    a.iborm[b] = lta ? min(2, a.smax) : a.smax;
    double tauout = H[0];
    if (a.jout) {
        int j = 0;
        double zz = 0.;
        if (a.zout != -1.0) {                                              // SOS.F:570-582, SOS_OS.F:1514-1520
            j = 1;
            while (j < nt && a.zout < Z[j]) j++;
            zz = (a.zout - Z[j - 1]) / (Z[j] - Z[j - 1]);
            tauout = (1 - zz) * H[j - 1] + zz * H[j];
        }
        a.jout[b] = j; a.zz[b] = zz;
    }
    a.scal[4 * b + 0] = 0.;
    a.scal[4 * b + 1] = H[nt];           // TTOT_TRONC
    a.scal[4 * b + 2] = ttot_vrai;       // TTOT_VRAI
    a.scal[4 * b + 3] = tauout;
}

void launch_profile(const ProfileArgs &a, hipStream_t st)
{
    // one wavefront per bin; SOSGPU_PROFILE_LANES=1: the one-lane-per-bin form (about 2048 wavefronts whatever the batch size)
    const char *e = getenv("SOSGPU_PROFILE_LANES");
    if (e && atoi(e)) {
        const int bpw = std::min(64, std::max(1, (a.nb + 2047) / 2048));
        k_profile<false><<<(a.nb + bpw - 1) / bpw, 64, 0, st>>>(a, bpw);
    } else k_profile<true><<<a.nb, 64, 0, st>>>(a, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// SOS_ABSPROFILE (src/SOS_ABSPROFILE.F:325-371) for every CKD bin of a wavelength: the absorption coefficient XK of a gas
// depends on the gas, on the exponential term chosen for it and on the layer only, so the host tabulates
// xk[gas][term][layer] (COEFF_ABS_CKD, SOS_SUB_TRS.F:171) once and a bin is one term index per gas.  Per bin, exactly the
// reference's loop: layer optical depth = sum over the eight gases (in order) of XK RO, transmission accumulated from the
// top, TAUABS(level) = -ln(TRS) or CTE_TAUABS_MAX when the transmission underflows.
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// One WAVEFRONT per bin (round 3; one thread per bin before: 49 dependent exp / log pairs, 80 us for a band -- a sixth of the
// device-side chain of a wavelength in sos_spectrum).  Lane j forms the optical depth of layer j (the eight gases added in the
// reference's order) and its transmission; the running product TRS is then formed in layer order -- the same left-to-right
// product as the loop's -- each lane keeping the value of its own layer, and takes its logarithm.  Bands with more layers than
// lanes walk the layers in blocks of 64 with the product carried over.
__global__ __launch_bounds__(64) void k_absprofile(int nb, int nlev, int nterm, const int32_t *__restrict__ ik,
                                                   const double *__restrict__ xk, const double *__restrict__ ro,
                                                   double *__restrict__ tabs)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= nb) return;
    const int nl1 = nlev - 1;
    int term[8];
    for (int k = 0; k < 8; k++) term[k] = min(max(ik[8 * b + k], 1), nterm) - 1;
    if (lane == 0) tabs[(size_t)b * nlev] = 0.;
    double trs = 1.0;                                            // TRS after the layers of the blocks before (wave-uniform)
    for (int j0 = 0; j0 < nl1; j0 += 64) {
        const int j = j0 + lane;
        double e = 1.0;
        if (j < nl1) {
            double t1c = 0.;
            for (int k = 0; k < 8; k++) t1c = t1c + xk[((size_t)k * nterm + term[k]) * nl1 + j] * ro[(size_t)k * nl1 + j];
            e = exp(-t1c);
        }
        double mine = 0.;
        const int cnt = min(64, nl1 - j0);
        for (int q = 0; q < cnt; q++) {                          // TRS = TRS * EXP(-T1C), layer by layer
            trs = trs * lane_read(e, q);
            if (q == lane) mine = trs;
        }
        if (j < nl1) tabs[(size_t)b * nlev + j + 1] = (mine > 0.) ? -log(mine) : 999.;           // CTE_TAUABS_MAX (SOS.h:297)
    }
}

void launch_absprofile(int nb, int nlev, int nterm, const int32_t *d_ik, const double *d_xk, const double *d_ro, double *d_tabs,
                       hipStream_t st)
{
    k_absprofile<<<nb, 64, 0, st>>>(nb, nlev, nterm, d_ik, d_xk, d_ro, d_tabs);
}
