// csrc/trphi.hip -- azimuth recomposition of the aggregated Fourier records (gfx950).
//
// Replaces SOS_TRPHI (reference src/SOS_TRPHI.F:749-1243) with SOS_GLITTE (:1278), SOS_ANGLE (:1347),
// SOS_REFLEX (:1433), SOS_MATRIC (:1505) and SOS_POLAR (:1843), for a whole list of azimuths in one
// launch (the reference re-opens and re-reads the result file once per azimuth, SOS_TRPHI.F:558-613).
// One workgroup per azimuth, one thread per direction jj in -N..N; the Fourier sum runs in the
// reference's order (s ascending).  HBM-streaming bound but tiny: F*3*(2N+1)*8 bytes per azimuth, L2 hits.
// The direct surface term covers the Cox-Munk glint (:946-1001), the flat-sea sun glint (:1008-1039) and the land models:
// Roujean (:1047-1076) and Rondeaux / Breon / Maignan (:1084-1136); Nadal (:1145-1200) is refused upstream by SOS_PROC.
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)
#include "land_models.h"

#define SEUIL_Z ((double)0.0001f)      // SOS.h:407 (REAL*4 literal)
#define SEUIL_X ((double)0.00001f)     // SOS.h:413
#define THRESHOLD_Q_U_NULL 1.e-15      // SOS.h:418
#define SOLAR_DISC_SOLID_ANGLE 6.8e-05 // SOS.h:426
#define VALEUR_INDEF (-999.)

__device__ inline void reflex(double cosdif, double ind, double &r11, double &r12, double &r33)
{   // SOS_TRPHI.F:1461-1470
    const double ind2 = ind * ind;
    const double cosw = sqrt(.5 * (1 - cosdif));
    const double v = .5 * (1 + cosdif);
    const double x = sqrt(ind2 - v);
    const double rl = (ind2 * cosw - x) / (ind2 * cosw + x);
    const double rr = (cosw - x) / (cosw + x);
    r11 = (rl * rl + rr * rr) / 2.;
    r12 = (rl * rl - rr * rr) / 2.;
    r33 = rr * rl;
}

// out[iphi][q][W], q = 0 XIT, 1 XQT, 2 XUT, 3 ANGDIFF, 4 XAN (polarisation angle), 5 TPOL, 6 LPOL
__global__ void k_trphi(SosDev cx, int nf, const double *__restrict__ rec, double tau, double tauout,
                        const double *__restrict__ phis, int igli, double sigma2, double ind_surf, LandTerms land,
                        double *__restrict__ out)
{
    const int iphi = blockIdx.x;
    const int N = cx.n, W = cx.w;
    const int t = threadIdx.x;
    if (t >= W) return;
    const int j = t - N;
    double *o = out + (size_t)iphi * 7 * W;
    if (j == 0) { for (int q = 0; q < 7; q++) o[q * W + t] = 0.; return; }
    const double pi = acos(-1.0);
    const double phi = phis[iphi];
    const double c0 = cx.mu[cx.n0 - 1];
    const double rmuj = j > 0 ? cx.mu[j - 1] : -cx.mu[-j - 1];
    // scattering angle, SOS_TRPHI.F:886-889
    const double cosd0 = -c0 * rmuj + sin(acos(c0)) * sin(acos(rmuj)) * cos(phi);
    const double angdiff = acos(cosd0) * 180.0 / pi;
    double xit = rec[0 * W + t], xqt = rec[1 * W + t], xut = rec[2 * W + t];
    for (int is = 1; is < nf; is++) {              // SOS_TRPHI.F:926-940
        const double *r = rec + (size_t)is * 3 * W;
        const double xphi = is * phi;
        const double cs = cos(xphi), sn = sin(xphi);
        xqt = xqt + 2. * r[1 * W + t] * cs;
        xut = xut + 2. * r[2 * W + t] * sn;
        xit = xit + 2. * r[0 * W + t] * cs;
    }
    // SOS_ANGLE (:1365-1372) + SOS_REFLEX + SOS_MATRIC (:1526-1538): first column of the Fresnel reflection matrix of the
    // facet that sends the sun (c0) into the direction c1 at azimuth phi
    auto fresnel_column = [&](double c1, double &m11, double &m21, double &m31, double &r12_out) {
        double s = 1.;
        if (sin(phi) > 0.0) s = -1.;
        const double cosdif = -c0 * c1 + sqrt(1 - c0 * c0) * sqrt(1 - c1 * c1) * cos(phi);
        const double z = s * (sqrt(1 - cosdif * cosdif)) * (sqrt(1 - c1 * c1));
        double coskip = 0.;
        if (fabs(z) > SEUIL_Z) coskip = (c1 * cosdif + c0) / z;
        double r11, r12, r33;
        reflex(cosdif, ind_surf, r11, r12, r33);
        const double x = 1. - fabs(coskip);
        double c2 = 1., s2 = 0.;
        if (x >= SEUIL_X) { c2 = 2. * coskip * coskip - 1.; s2 = 2. * coskip * sqrt(1. - coskip * coskip); }
        if (coskip == 0.0) r12 = 0.;
        m11 = r11; m21 = c2 * r12; m31 = s2 * r12; r12_out = r12;
    };
    if (igli == 1 && j > 0) {                      // SOS_TRPHI.F:946-1001
        const double c1 = rmuj;
        const double at0 = exp(-tau / c0);
        const double atj = at0 * exp(-(tau - tauout) / c1);
        // SOS_GLITTE :1303-1314
        double p;
        {
            const double x1 = sqrt(1 - c1 * c1) - cos(phi) * sqrt(1 - c0 * c0);
            const double x2 = sqrt(1 - c0 * c0) * sin(phi);
            const double x3 = c0 + c1;
            const double c0n = (x3 / (sqrt(x1 * x1 + x2 * x2 + x3 * x3)));
            const double xxx = (-(1 - c0n * c0n) / (sigma2 * c0n * c0n));
            if (xxx < -100) p = 0.;
            else {
                const double pp = (1 / sigma2) * exp(xxx);
                const double c0n2 = c0n * c0n;
                p = pp / (4 * c1 * (c0n2 * c0n2));
            }
        }
        double m11, m21, m31, r12;
        fresnel_column(c1, m11, m21, m31, r12);
        xit = xit + m11 * atj * p;
        if (cx.ipolar == 1) { xqt = xqt + m21 * atj * p; xut = xut + m31 * atj * p; }
    }
    if (land.iroujean == 1 && j > 0) {             // SOS_TRPHI.F:1047-1076
        const double c1 = rmuj;
        const double atj = exp(-tau / c0) * exp(-(tau - tauout) / c1);
        const double f = calc_f_roujean(land.k0, land.k1, land.k2, c0, sqrt(1. - c0 * c0), c1, sqrt(1. - c1 * c1), pi - phi);
        xit = xit + atj * f / c1;
    }
    if ((land.irondeaux == 1 || land.ibreon == 1 || land.imaignan == 1) && j > 0) {   // SOS_TRPHI.F:1084-1136
        const double c1 = rmuj;
        const double atj = exp(-tau / c0) * exp(-(tau - tauout) / c1);
        double m11, m21, m31, r12, p = 0.;
        fresnel_column(c1, m11, m21, m31, r12);
        if (land.irondeaux == 1) p = 1. / (4. * (1 + c1 / c0));
        if (land.ibreon == 1) p = 1. / (4. * c1);
        if (land.imaignan == 1) p = calcg_maignan(c0, c1, sqrt(1. - c0 * c0) * sqrt(1. - c1 * c1), phi, land.coef_c) / (4. * c1);
        xit = xit + m11 * atj * p;
        if (cx.ipolar == 1) { xqt = xqt + m21 * atj * p; xut = xut + m31 * atj * p; }
    }
    if (cx.ifresnel == 1 && j == cx.n0) {          // SOS_TRPHI.F:1008-1039
        if (cos(phi) == 1.0) {
            const double at0 = exp(-tau / c0);
            const double atj = at0 * exp(-(tau - tauout) / c0);
            const double cosdif = 1. - 2. * c0 * c0;
            double r11, r12, r33;
            reflex(cosdif, ind_surf, r11, r12, r33);
            const double coef_sun = pi / SOLAR_DISC_SOLID_ANGLE;
            xit = xit + r11 * coef_sun * atj;
            if (cx.ipolar == 1) xqt = xqt + r12 * coef_sun * atj;
        }
    }
    if (xit <= 1.e-99) xit = 0.0;                   // SOS_TRPHI.F:1212-1218
    if (fabs(xqt) < THRESHOLD_Q_U_NULL) xqt = 0.0;
    if (fabs(xut) < THRESHOLD_Q_U_NULL) xut = 0.0;
    // SOS_POLAR :1865-1903
    double xan, tpol, lpol;
    if (xqt != 0.) {
        const double xt = xut / xqt;
        if (xqt > 0.) xan = 90. * atan(xt) / pi;
        else if (xut > 0.) xan = 90. + 90. * atan(xt) / pi;
        else xan = -90. + 90. * atan(xt) / pi;
    } else {
        if (xut > 0.) xan = 45.;
        else if (xut < 0) xan = -45.;
        else xan = VALEUR_INDEF;
    }
    lpol = sqrt(xqt * xqt + xut * xut);
    tpol = (xit != 0.0) ? 100. * lpol / xit : VALEUR_INDEF;
    o[0 * W + t] = xit; o[1 * W + t] = xqt; o[2 * W + t] = xut; o[3 * W + t] = angdiff;
    o[4 * W + t] = xan; o[5 * W + t] = tpol; o[6 * W + t] = lpol;
}

void launch_trphi(const SosDev &cx, int nf, const double *d_rec, double tau, double tauout, int nphi,
                  const double *d_phi, int igli, double sigma2, double ind_surf, const LandTerms &land, double *d_out,
                  hipStream_t st)
{
    k_trphi<<<nphi, sos_round_up(cx.w, 64), 0, st>>>(cx, nf, d_rec, tau, tauout, d_phi, igli, sigma2, ind_surf, land, d_out);
}
