/* oracle/sos_trphi_oracle.c -- CPU restatement of the azimuth recomposition.
 *
 * TEST INFRASTRUCTURE ONLY (see sos_oracle.h).
 *
 * Follows SOS_TRPHI (src/SOS_TRPHI.F:749-1243) with SOS_GLITTE (:1278), SOS_ANGLE (:1347), SOS_REFLEX
 * (:1433), SOS_MATRIC (:1505) and SOS_POLAR (:1843).  Land BRDF/BPDF direct terms (Roujean, Rondeaux,
 * Breon, Nadal, Maignan; :1047-1200) are outside the round-1 scope.
 */
#include "sos_oracle.h"
#include <math.h>
#include <stdlib.h>

#define SEUIL_Z ((double)0.0001f)   /* SOS.h:407, REAL*4 literal */
#define SEUIL_X ((double)0.00001f)  /* SOS.h:413 */
#define THRESHOLD_Q_U_NULL 1.e-15   /* SOS.h:418 */
#define SOLAR_DISC_SOLID_ANGLE 6.8e-05 /* SOS.h:426 */
#define VALEUR_INDEF (-999.)

double sos_oracle_sigma2(double wind);

static void glitte(double sig, double c0, double c1, double phi, double *p) /* :1303-1314 */
{
    double x1 = sqrt(1 - c1 * c1) - cos(phi) * sqrt(1 - c0 * c0);
    double x2 = sqrt(1 - c0 * c0) * sin(phi);
    double x3 = c0 + c1;
    double c0n = (x3 / (sqrt(x1 * x1 + x2 * x2 + x3 * x3)));
    double xxx = (-(1 - c0n * c0n) / (sig * c0n * c0n));
    if (xxx < -100) *p = 0.;
    else {
        double pp = (1 / sig) * exp(xxx);
        double c0n2 = c0n * c0n;
        *p = pp / (4 * c1 * (c0n2 * c0n2)); /* C0N**4 by repeated squaring */
    }
}

static void angle(double c0, double c1, double phi, double *coskip, double *cosdif) /* :1365-1372 */
{
    double s = 1., z;
    if (sin(phi) > 0.0) s = -1.;
    *cosdif = -c0 * c1 + sqrt(1 - c0 * c0) * sqrt(1 - c1 * c1) * cos(phi);
    z = s * (sqrt(1 - (*cosdif) * (*cosdif))) * (sqrt(1 - c1 * c1));
    *coskip = 0.;
    if (fabs(z) > SEUIL_Z) *coskip = (c1 * (*cosdif) + c0) / z;
}

static void reflex(double cosdif, double ind, double *r11, double *r12, double *r33) /* :1461-1470 */
{
    double ind2 = ind * ind;
    double cosw = sqrt(.5 * (1 - cosdif));
    double v = .5 * (1 + cosdif);
    double x = sqrt(ind2 - v);
    double rl = (ind2 * cosw - x) / (ind2 * cosw + x);
    double rr = (cosw - x) / (cosw + x);
    *r11 = (rl * rl + rr * rr) / 2.;
    *r12 = (rl * rl - rr * rr) / 2.;
    *r33 = rr * rl;
}

static void matric(double coskip, double r11, double *r12, double *m11, double *m21, double *m31) /* :1526-1538 */
{
    double x = 1. - fabs(coskip), c2 = 1., s2 = 0.;
    if (x >= SEUIL_X) {
        c2 = 2. * coskip * coskip - 1.;
        s2 = 2. * coskip * sqrt(1. - coskip * coskip);
    }
    if (coskip == 0.0) *r12 = 0.;
    *m11 = r11;
    *m21 = c2 * (*r12);
    *m31 = s2 * (*r12);
}

/* SOS_POLAR :1865-1903 */
void sos_oracle_polar(double xi, double xq, double xu, double *xan, double *tpol, double *lpol)
{
    const double pi = acos(-1.0);
    if (xq != 0.) {
        double xt = xu / xq;
        if (xq > 0.) *xan = 90. * atan(xt) / pi;
        else if (xu > 0.) *xan = 90. + 90. * atan(xt) / pi;
        else *xan = -90. + 90. * atan(xt) / pi;
    } else {
        if (xu > 0.) *xan = 45.;
        else if (xu < 0) *xan = -45.;
        else *xan = VALEUR_INDEF;
    }
    *lpol = sqrt(xq * xq + xu * xu);
    if (xi != 0.0) *tpol = 100. * (*lpol) / xi;
    else *tpol = VALEUR_INDEF;
}

/* SOS_TRPHI for one azimuth phi (radians).
 *  rec [nf][3][W] aggregated Fourier records (I,Q,U); mu[n]; n0 (1-based solar index)
 *  outputs xit,xqt,xut,angdiff [W] (slot jj = 0: angdiff as the reference computes it from RMU(0)=0
 *  is not reproduced -- set to 0). */
void sos_oracle_trphi(int n, const double *mu, int nf, const double *rec, double tau, double tauout, double phi,
                      int igli, int n0, double wind, double ind_surf, int ifresnel, int ipolar,
                      double *xit, double *xqt, double *xut, double *angdiff)
{
    const double pi = acos(-1.0);
    const int W = 2 * n + 1;
    int j, is;
    double c0 = mu[n0 - 1];
#define RMU(j) ((j) > 0 ? mu[(j)-1] : -mu[-(j)-1])
    for (j = -n; j <= n; j++) { /* :884-891 */
        double cosdif;
        if (j == 0) { angdiff[n] = 0.; continue; }
        cosdif = -c0 * RMU(j) + sin(acos(c0)) * sin(acos(RMU(j))) * cos(phi);
        angdiff[j + n] = acos(cosdif) * 180.0 / pi;
    }
    for (j = 0; j < W; j++) { xit[j] = rec[0 * W + j]; xqt[j] = rec[1 * W + j]; xut[j] = rec[2 * W + j]; }
    xit[n] = xqt[n] = xut[n] = 0.; /* slot jj = 0 is never assigned by the reference (:913-918) */
    for (is = 1; is < nf; is++) { /* :926-940 */
        const double *r = rec + (size_t)is * 3 * W;
        double xphi = is * phi;
        for (j = -n; j <= n; j++) {
            if (j == 0) continue;
            xqt[j + n] = xqt[j + n] + 2. * r[1 * W + j + n] * cos(xphi);
            xut[j + n] = xut[j + n] + 2. * r[2 * W + j + n] * sin(xphi);
            xit[j + n] = xit[j + n] + 2. * r[0 * W + j + n] * cos(xphi);
        }
    }
    if (igli == 1) { /* :946-1001 */
        double at0 = exp(-tau / c0);
        double sigma2 = sos_oracle_sigma2(wind);
        for (j = 1; j <= n; j++) {
            double c1 = mu[j - 1], p, coskip, cosdif, r11, r12, r33, m11, m21, m31;
            double atj = at0 * exp(-(tau - tauout) / c1);
            glitte(sigma2, c0, c1, phi, &p);
            angle(c0, c1, phi, &coskip, &cosdif);
            reflex(cosdif, ind_surf, &r11, &r12, &r33);
            matric(coskip, r11, &r12, &m11, &m21, &m31);
            xit[j + n] = xit[j + n] + m11 * atj * p;
            if (ipolar == 1) {
                xqt[j + n] = xqt[j + n] + m21 * atj * p;
                xut[j + n] = xut[j + n] + m31 * atj * p;
            }
        }
    }
    if (ifresnel == 1) { /* :1008-1039 */
        if ((cos(phi) == 1.0) && (n0 > 0)) {
            double at0 = exp(-tau / c0);
            double atj = at0 * exp(-(tau - tauout) / c0);
            double cosdif = 1. - 2. * c0 * c0, r11, r12, r33;
            double coef_sun = pi / SOLAR_DISC_SOLID_ANGLE;
            reflex(cosdif, ind_surf, &r11, &r12, &r33);
            xit[n0 + n] = xit[n0 + n] + r11 * coef_sun * atj;
            if (ipolar == 1) xqt[n0 + n] = xqt[n0 + n] + r12 * coef_sun * atj;
        }
    }
    for (j = -n; j <= n; j++) { /* :1212-1218 */
        if (xit[j + n] <= 1.e-99) xit[j + n] = 0.0;
        if (fabs(xqt[j + n]) < THRESHOLD_Q_U_NULL) xqt[j + n] = 0.0;
        if (fabs(xut[j + n]) < THRESHOLD_Q_U_NULL) xut[j + n] = 0.0;
    }
#undef RMU
}
