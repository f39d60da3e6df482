/* oracle/sos_oracle.c -- CPU restatement (plain C, fp64) of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sos_oracle.h).  Never shipped, never on the product path.
 *
 * Each function follows the reference's operation order (file:line cited) so that data-dependent
 * stop decisions agree; this includes the places where the Fortran evaluates a sub-expression in
 * REAL*4 before widening (marked "REAL*4").  Compile with -ffp-contract=off.
 */
#include "sos_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* inc/SOS.h:389,394,400 are REAL*4 literals (0.00001) compared against DOUBLE PRECISION Z1;
 * SOS.h:395 is a D literal. */
#define SEUIL_CV_SG  ((double)0.00001f)
#define SEUIL_SUMDIF ((double)0.00001f)
#define SEUIL_VALDIF 1.0e-50
#define SEUIL_SF     ((double)0.00001f)
#define TOA_ALT      120.0 /* SOS.h:197 */

/* ------------------------------------------------------------------------------------------- */
/* SOS_NOYAUX  (SOS_OS.F:1857-2158)                                                              */
/* ------------------------------------------------------------------------------------------- */
void sos_oracle_noyaux(int is, int n, double rmu0, const double *mu, int os_nb,
                       const double *alpha, const double *beta, const double *gamma, const double *zeta,
                       double *xpl, double *xrl, double *xtl,
                       double *bp, double *gr, double *gt, double *arr, double *art, double *att)
{
    const int W = 2 * n + 1;
    const int NL = os_nb + 2; /* l = -1 .. os_nb */
    double *psl = calloc((size_t)NL * W, sizeof(double));
    double *rsl = calloc((size_t)NL * W, sizeof(double));
    double *tsl = calloc((size_t)NL * W, sizeof(double));
    /* RSL(0,.) and TSL(0,.) are never assigned by the reference for IS=0 (stack garbage multiplied by
     * alpha(0)=zeta(0)=0 in physical inputs); calloc gives the intended 0. */
#define P(l, j) psl[((j) + n) * NL + (l) + 1]
#define R(l, j) rsl[((j) + n) * NL + (l) + 1]
#define T(l, j) tsl[((j) + n) * NL + (l) + 1]
#define RMU(j) ((j) == 0 ? rmu0 : ((j) > 0 ? mu[(j)-1] : -mu[-(j)-1]))
    const double rac3 = sqrt(3.0);
    const double x26 = 2. * sqrt(6.0);
    int j, k, l;

    if (is == 0) { /* :1970-1991 */
        for (j = 0; j <= n; j++) {
            double c = RMU(j), x;
            P(0, -j) = 1.; P(0, j) = 1.;
            P(1, j) = c;   P(1, -j) = -c;
            x = (3. * c * c - 1.) * 0.5;
            P(2, -j) = x; P(2, j) = x;
            R(1, j) = 0.; R(1, -j) = 0.;
            x = 3. * (1. - c * c) / x26;
            R(2, -j) = x; R(2, j) = x;
            T(1, j) = 0.; T(1, -j) = 0.; T(2, j) = 0.; T(2, -j) = 0.;
        }
        P(1, 0) = RMU(0);
        R(1, 0) = 0.;
    } else if (is == 1) { /* :1999-2021 */
        for (j = 0; j <= n; j++) {
            double c = RMU(j), x = 1. - c * c;
            P(0, j) = 0.; P(0, -j) = 0.;
            P(1, -j) = sqrt(x * 0.5);
            P(1, j) = sqrt(x * 0.5);
            P(2, j) = c * P(1, j) * rac3;
            P(2, -j) = -P(2, j);
            R(1, -j) = 0.; R(1, j) = 0.;
            R(2, j) = -c * sqrt(x) * 0.5;
            R(2, -j) = -R(2, j);
            T(1, -j) = 0.; T(1, j) = 0.;
            T(2, j) = -sqrt(x) * 0.5;
            T(2, -j) = -sqrt(x) * 0.5;
        }
        P(2, 0) = -P(2, 0);
        R(2, 0) = -R(2, 0);
        R(1, 0) = 0.;
        T(1, 0) = 0.;
    } else { /* :2027-2052 */
        double a = 1., b;
        int i;
        for (i = 1; i <= is; i++) {
            double x = i;
            a = a * sqrt((i + is) / x) * 0.5;
        }
        b = a * sqrt(is / (is + 1.0)) * sqrt((is - 1.0) / (is + 2.));
        for (j = 0; j <= n; j++) {
            double c = RMU(j), xx = 1. - c * c, yy = is * 0.5 - 1., x;
            P(is - 1, j) = 0.; R(is - 1, j) = 0.; T(is - 1, j) = 0.;
            x = a * pow(xx, is * 0.5);
            P(is, -j) = x; P(is, j) = x;
            x = b * (1. + c * c) * pow(xx, yy);
            R(is, -j) = x; R(is, j) = x;
            x = 2. * b * c * pow(xx, yy);
            T(is, -j) = -x; T(is, j) = x;
        }
    }
    /* recurrence :2058-2100 */
    k = 2;
    if (is > 2) k = is;
    if (k != os_nb) {
        int ig = -1;
        if (is == 1) ig = 1;
        for (l = k; l <= os_nb - 1; l++) {
            int lp = l + 1, lm = l - 1;
            double a = (2 * l + 1.) / sqrt((l + is + 1.0) * (l - is + 1.));
            double b = sqrt((double)((l + is) * (l - is))) / (2. * l + 1.);
            double d = (l + 1.) * (2 * l + 1.) / sqrt((l + 3.0) * (l - 1.) * (l + is + 1.) * (l - is + 1.));
            double e = sqrt((l + 2.0) * (l - 2.) * (l + is) * (l - is)) / (l * (2. * l + 1.));
            /* REAL*4: F=2.*IS/(L*(L+1.)) has only REAL*4/INTEGER operands (:2079) */
            double f = (double)((2.f * (float)is) / ((float)l * ((float)l + 1.f)));
            for (j = 0; j <= n; j++) {
                double c = RMU(j), x;
                x = a * (c * P(l, j) - b * P(lm, j));
                P(lp, j) = x;
                x = d * (c * R(l, j) - f * T(l, j) - e * R(lm, j));
                R(lp, j) = x;
                x = d * (c * T(l, j) - f * R(l, j) - e * T(lm, j));
                T(lp, j) = x;
                if (j == 0) continue;
                P(lp, -j) = ig * P(lp, j);
                R(lp, -j) = ig * R(lp, j);
                T(lp, -j) = -ig * T(lp, j);
            }
            ig = -ig;
        }
    }
    for (j = -n; j <= n; j++) { /* :2107-2111 */
        xpl[j + n] = P(2, j);
        xrl[j + n] = R(2, j);
        xtl[j + n] = T(2, j);
    }
    for (j = -n; j <= n; j++) { /* :2121-2155 */
        for (k = -n; k <= n; k++) {
            double sbp = 0., satt = 0., sarr = 0., sgr = 0., sgt = 0., sart = 0.;
            if (is <= os_nb) {
                for (l = is; l <= os_nb; l++) {
                    double r1 = T(l, j) * T(l, k);
                    double r2 = R(l, j) * R(l, k);
                    sbp = sbp + beta[l] * P(l, j) * P(l, k);
                    satt = satt + alpha[l] * r1 + zeta[l] * r2;
                    sarr = sarr + zeta[l] * r1 + alpha[l] * r2;
                    sgr = sgr + gamma[l] * P(l, j) * R(l, k);
                    sgt = sgt + gamma[l] * P(l, j) * T(l, k);
                    sart = sart + alpha[l] * R(l, k) * T(l, j) + zeta[l] * R(l, j) * T(l, k);
                }
            }
            size_t o = (size_t)(j + n) * W + (k + n);
            bp[o] = sbp; att[o] = satt; arr[o] = sarr; gr[o] = sgr; gt[o] = sgt; art[o] = sart;
        }
    }
#undef P
#undef R
#undef T
#undef RMU
    free(psl); free(rsl); free(tsl);
}

/* ------------------------------------------------------------------------------------------- */
/* SOS_OS and its leaves                                                                         */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int n, nt, L, W;
    double *rmu; /* W, RMU(-N:N) incl. RMU(0) = -mus */
    double *ga;  /* W */
} geom_t;

#define FLD(a, i, k) (a)[(size_t)((k) + g->n) * g->L + (i)]
#define KER(a, j, k) (a)[(size_t)((j) + g->n) * g->W + ((k) + g->n)]
#define V(a, k) (a)[(k) + g->n]

/* SOS_INTEGR_EPOPT, SOS_OS.F:2222-2357 */
static void integr_epopt(const geom_t *g, const double *h, const double *i2, const double *q2, const double *u2,
                         double *i1, double *q1, double *u1)
{
    int n = g->n, nt = g->nt, i, k;
    for (k = 1; k <= n; k++) {
        double rmuk = V(g->rmu, k);
        double zi1 = FLD(i1, nt, k), zq1 = FLD(q1, nt, k), zu1 = FLD(u1, nt, k);
        for (i = nt - 1; i >= 0; i--) {
            int jj = i + 1;
            double dtau = h[jj] - h[i];
            double att = exp(-dtau / rmuk);
            double matt = 1.0 - att;
            double attdtau = att * dtau;
            double b, a;
            b = FLD(i2, i, k); a = (FLD(i2, jj, k) - b) / dtau;
            zi1 = zi1 * att + matt * (a * rmuk + b) - a * attdtau;
            FLD(i1, i, k) = zi1;
            b = FLD(q2, i, k); a = (FLD(q2, jj, k) - b) / dtau;
            zq1 = zq1 * att + matt * (a * rmuk + b) - a * attdtau;
            FLD(q1, i, k) = zq1;
            b = FLD(u2, i, k); a = (FLD(u2, jj, k) - b) / dtau;
            zu1 = zu1 * att + matt * (a * rmuk + b) - a * attdtau;
            FLD(u1, i, k) = zu1;
        }
    }
    for (k = -n; k <= -1; k++) {
        double rmuk = V(g->rmu, k);
        double zi1 = 0., zq1 = 0., zu1 = 0.;
        FLD(i1, 0, k) = 0.; FLD(q1, 0, k) = 0.; FLD(u1, 0, k) = 0.;
        for (i = 1; i <= nt; i++) {
            int jj = i - 1;
            double dtau = h[i] - h[jj];
            double att = exp(dtau / rmuk);
            double matt = 1.0 - att;
            double attdtau = att * dtau;
            double b, a;
            b = FLD(i2, i, k); a = (b - FLD(i2, jj, k)) / dtau;
            zi1 = zi1 * att + matt * (a * rmuk + b) + a * attdtau;
            FLD(i1, i, k) = zi1;
            b = FLD(q2, i, k); a = (b - FLD(q2, jj, k)) / dtau;
            zq1 = zq1 * att + matt * (a * rmuk + b) + a * attdtau;
            FLD(q1, i, k) = zq1;
            b = FLD(u2, i, k); a = (b - FLD(u2, jj, k)) / dtau;
            zu1 = zu1 * att + matt * (a * rmuk + b) + a * attdtau;
            FLD(u1, i, k) = zu1;
        }
    }
}

typedef struct {
    double *xpl, *xrl, *xtl;               /* W */
    double *bp, *gr, *gt, *arr, *art, *att; /* W*W */
} kern_t;

/* SOS_FSOURCE_ORDRE1, SOS_OS.F:2431-2565 (JK = 0) */
static void fsource_ordre1(const geom_t *g, int is, const double *xdel, const double *ydel,
                           double beta0, double beta2, double gamma2, const kern_t *kn, const double *ch,
                           double *i2, double *q2, double *u2)
{
    int n = g->n, nt = g->nt, j, k;
    for (j = -n; j <= n; j++) {
        double sa1, sa2, sb1, sb2, sc1, sc2;
        if (is - 2 > 0) {
            sa2 = KER(kn->bp, 0, j); sa1 = 0.;
            sb2 = KER(kn->gr, 0, j); sb1 = 0.;
            sc2 = KER(kn->gt, 0, j); sc1 = 0.;
        } else {
            double spl = V(kn->xpl, 0);
            sa1 = beta0 + beta2 * V(kn->xpl, j) * spl;
            sa2 = KER(kn->bp, 0, j);
            sb1 = gamma2 * V(kn->xrl, j) * spl;
            sb2 = KER(kn->gr, 0, j);
            sc1 = gamma2 * V(kn->xtl, j) * spl;
            sc2 = KER(kn->gt, 0, j);
        }
        for (k = 0; k <= nt; k++) {
            double attdir = ch[k], pcray = ydel[k], pcaer = xdel[k];
            FLD(i2, k, j) = attdir * (sa2 * pcaer + sa1 * pcray);
            FLD(q2, k, j) = attdir * (sb2 * pcaer + sb1 * pcray);
            FLD(u2, k, j) = -attdir * (sc2 * pcaer + sc1 * pcray);
        }
    }
}

/* SOS_FSOURCE_ORDREIG, SOS_OS.F:2663-3017 */
static void fsource_ordreig(const geom_t *g, int is, const double *xdel, const double *ydel,
                            double beta0, double beta2, double gamma2, double alpha2, const kern_t *kn,
                            const double *i1, const double *q1, const double *u1,
                            double *i2, double *q2, double *u2)
{
    int n = g->n, nt = g->nt, i, j, k;
    const int ray = !(is - 2 > 0);
    for (k = 1; k <= n; k++) {
        double xpk = V(kn->xpl, k), xrk = V(kn->xrl, k), xtk = V(kn->xtl, k);
        double ypk = V(kn->xpl, -k), yrk = V(kn->xrl, -k), ytk = V(kn->xtl, -k);
        for (i = 0; i <= nt; i++) {
            double ii1 = 0., ii2 = 0., qq1 = 0., qq2 = 0., uu1 = 0., uu2 = 0.;
            double pcaer = xdel[i], pcray = ydel[i];
            for (j = 1; j <= n; j++) {
                double bpjk, bpjmk, gtjmk, gtjk, gtkmj, gtkj, grjk, grjmk, grkj, grkmj;
                double arrjk, arrjmk, artjk, artjmk, artkj, artkmj, attjmk, attjk;
                if (ray) { /* :2852-2876 */
                    double xpj = V(kn->xpl, j), xrj = V(kn->xrl, j), xtj = V(kn->xtl, j);
                    double yrj = V(kn->xrl, -j), ytj = V(kn->xtl, -j);
                    bpjk = KER(kn->bp, j, k) * pcaer + pcray * (beta0 + beta2 * xpj * xpk);
                    bpjmk = KER(kn->bp, j, -k) * pcaer + pcray * (beta0 + beta2 * xpj * ypk);
                    gtjmk = KER(kn->gt, j, -k) * pcaer + pcray * (gamma2 * xpj * ytk);
                    gtjk = KER(kn->gt, j, k) * pcaer + pcray * (gamma2 * xpj * xtk);
                    gtkmj = KER(kn->gt, k, -j) * pcaer + pcray * (gamma2 * xpk * ytj);
                    gtkj = KER(kn->gt, k, j) * pcaer + pcray * (gamma2 * xpk * xtj);
                    grjk = KER(kn->gr, j, k) * pcaer + pcray * (gamma2 * xpj * xrk);
                    grjmk = KER(kn->gr, j, -k) * pcaer + pcray * (gamma2 * xpj * yrk);
                    grkj = KER(kn->gr, k, j) * pcaer + pcray * (gamma2 * xpk * xrj);
                    grkmj = KER(kn->gr, k, -j) * pcaer + pcray * (gamma2 * xpk * yrj);
                    arrjk = KER(kn->arr, j, k) * pcaer + pcray * (alpha2 * xrj * xrk);
                    arrjmk = KER(kn->arr, j, -k) * pcaer + pcray * (alpha2 * xrj * yrk);
                    artjk = KER(kn->art, j, k) * pcaer + pcray * (alpha2 * xtj * xrk);
                    artjmk = KER(kn->art, j, -k) * pcaer + pcray * (alpha2 * xtj * yrk);
                    artkj = KER(kn->art, k, j) * pcaer + pcray * (alpha2 * xtk * xrj);
                    artkmj = KER(kn->art, k, -j) * pcaer + pcray * (alpha2 * xtk * yrj);
                    attjmk = KER(kn->att, j, -k) * pcaer + pcray * (alpha2 * xtj * ytk);
                    attjk = KER(kn->att, j, k) * pcaer + pcray * (alpha2 * xtj * xtk);
                } else { /* :2951-2968 */
                    bpjk = KER(kn->bp, j, k) * pcaer;     bpjmk = KER(kn->bp, j, -k) * pcaer;
                    gtjmk = KER(kn->gt, j, -k) * pcaer;   gtjk = KER(kn->gt, j, k) * pcaer;
                    gtkmj = KER(kn->gt, k, -j) * pcaer;   gtkj = KER(kn->gt, k, j) * pcaer;
                    grjk = KER(kn->gr, j, k) * pcaer;     grjmk = KER(kn->gr, j, -k) * pcaer;
                    grkj = KER(kn->gr, k, j) * pcaer;     grkmj = KER(kn->gr, k, -j) * pcaer;
                    arrjk = KER(kn->arr, j, k) * pcaer;   arrjmk = KER(kn->arr, j, -k) * pcaer;
                    artjk = KER(kn->art, j, k) * pcaer;   artjmk = KER(kn->art, j, -k) * pcaer;
                    artkj = KER(kn->art, k, j) * pcaer;   artkmj = KER(kn->art, k, -j) * pcaer;
                    attjmk = KER(kn->att, j, -k) * pcaer; attjk = KER(kn->att, j, k) * pcaer;
                }
                {
                    double z = V(g->ga, j);
                    double xi1 = FLD(i1, i, j), xi2 = FLD(i1, i, -j);
                    double xq1 = FLD(q1, i, j), xq2 = FLD(q1, i, -j);
                    double xu1 = FLD(u1, i, j), xu2 = FLD(u1, i, -j);
                    /* :2894-2905 */
                    ii2 = ii2 + z * (xi1 * bpjk + xi2 * bpjmk + xq1 * grkj + xq2 * grkmj - xu1 * gtkj - xu2 * gtkmj);
                    ii1 = ii1 + z * (xi1 * bpjmk + xi2 * bpjk + xq1 * grkmj + xq2 * grkj + xu1 * gtkmj + xu2 * gtkj);
                    qq2 = qq2 + z * (xi1 * grjk + xi2 * grjmk + xq1 * arrjk + xq2 * arrjmk + xu2 * artjmk - xu1 * artjk);
                    qq1 = qq1 + z * (xi1 * grjmk + xi2 * grjk + xq1 * arrjmk + xq2 * arrjk - xu1 * artjmk + xu2 * artjk);
                    uu2 = uu2 - z * (xi1 * gtjk - xi2 * gtjmk + xq1 * artkj + xq2 * artkmj - xu1 * attjk - xu2 * attjmk);
                    uu1 = uu1 - z * (xi1 * gtjmk - xi2 * gtjk - xq1 * artkmj - xq2 * artkj - xu1 * attjmk - xu2 * attjk);
                }
            }
            FLD(i2, i, k) = ii2 * 0.5; FLD(i2, i, -k) = ii1 * 0.5;
            FLD(q2, i, k) = qq2 * 0.5; FLD(q2, i, -k) = qq1 * 0.5;
            FLD(u2, i, k) = uu2 * 0.5; FLD(u2, i, -k) = uu1 * 0.5;
        }
    }
}

/* SOS_FSOURCE_DIFF_FRESNEL1, SOS_OS.F:3106-3295 */
static void fsource_diff_fresnel1(const geom_t *g, int is, double f11sun, double f12sun,
                                  const double *xdel, const double *ydel,
                                  double beta0, double beta2, double gamma2, double alpha2, const kern_t *kn,
                                  double mus, const double *h, double *i2, double *q2, double *u2)
{
    int n = g->n, nt = g->nt, j, k;
    memset(i2, 0, sizeof(double) * g->L * g->W);
    memset(q2, 0, sizeof(double) * g->L * g->W);
    memset(u2, 0, sizeof(double) * g->L * g->W);
    double coefnt = exp(2. * h[nt] / mus) / 4.;
    double spl = V(kn->xpl, 0);
    for (k = 0; k <= nt - 1; k++) {
        double yr = ydel[k], xp = xdel[k], yyr = ydel[k + 1], xxp = xdel[k + 1];
        for (j = 1; j <= n; j++) {
            double bp0mj, bp0j, grj0, gr0j, gr0mj, grmj0, gt0mj, gt0j, arr0mj, arr0j, artj0, artmj0;
            if (is <= 2) { /* :3237-3252 */
                bp0mj = KER(kn->bp, 0, -j) * xp + (beta0 + beta2 * V(kn->xpl, -j) * spl) * yr;
                bp0j = KER(kn->bp, 0, j) * xxp + (beta0 + beta2 * V(kn->xpl, j) * spl) * yyr;
                grj0 = KER(kn->gr, j, 0) * xxp + yyr * V(kn->xrl, 0) * V(kn->xpl, j) * gamma2;
                gr0j = KER(kn->gr, 0, j) * xxp + yyr * V(kn->xrl, j) * V(kn->xpl, 0) * gamma2;
                gr0mj = KER(kn->gr, 0, -j) * xp + yr * V(kn->xrl, -j) * spl * gamma2;
                grmj0 = KER(kn->gr, -j, 0) * xp + yr * gamma2 * V(kn->xrl, 0) * V(kn->xpl, -j);
                gt0mj = KER(kn->gt, 0, -j) * xp + yr * gamma2 * spl * V(kn->xtl, -j);
                gt0j = KER(kn->gt, 0, j) * xxp + yyr * gamma2 * spl * V(kn->xtl, j);
                arr0mj = KER(kn->arr, 0, -j) * xp + alpha2 * yr * V(kn->xrl, 0) * V(kn->xrl, -j);
                arr0j = KER(kn->arr, 0, j) * xxp + alpha2 * yyr * V(kn->xrl, 0) * V(kn->xrl, j);
                artj0 = KER(kn->art, j, 0) * xxp + yyr * alpha2 * V(kn->xtl, j) * V(kn->xrl, 0);
                artmj0 = KER(kn->art, -j, 0) * xp + yr * alpha2 * V(kn->xtl, -j) * V(kn->xrl, 0);
            } else {
                bp0mj = KER(kn->bp, 0, -j) * xp;   bp0j = KER(kn->bp, 0, j) * xxp;
                grj0 = KER(kn->gr, j, 0) * xxp;    gr0j = KER(kn->gr, 0, j) * xxp;
                gr0mj = KER(kn->gr, 0, -j) * xp;   grmj0 = KER(kn->gr, -j, 0) * xp;
                gt0mj = KER(kn->gt, 0, -j) * xp;   gt0j = KER(kn->gt, 0, j) * xxp;
                arr0mj = KER(kn->arr, 0, -j) * xp; arr0j = KER(kn->arr, 0, j) * xxp;
                artj0 = KER(kn->art, j, 0) * xxp;  artmj0 = KER(kn->art, -j, 0) * xp;
            }
            {
                double coefk = coefnt * exp(-h[k] / mus);
                double coefkp1;
                FLD(i2, k, j) = coefk * (f11sun * bp0mj + f12sun * grmj0);
                FLD(q2, k, j) = coefk * (f11sun * gr0mj + f12sun * arr0mj);
                FLD(u2, k, j) = coefk * (f11sun * gt0mj + f12sun * artmj0);
                coefkp1 = coefnt * exp(-h[k + 1] / mus);
                FLD(i2, k + 1, -j) = coefkp1 * (f11sun * bp0j + f12sun * grj0);
                FLD(q2, k + 1, -j) = coefkp1 * (f11sun * gr0j + f12sun * arr0j);
                FLD(u2, k + 1, -j) = coefkp1 * (f11sun * gt0j + f12sun * artj0);
            }
        }
    }
}

/* one term of SOS_PARAM_CONV, SOS_OS.F:3434-3453 */
/* Tie audit ("near-threshold ties need a documented policy"): every stop decision of the last
 * sos_oracle_os call records how far the tested value was from its threshold, |Z1/threshold - 1|; the minimum is read
 * with sos_oracle_stop_margin().  A decision closer to its threshold than the rounding difference between two correct
 * implementations (about 1e-12) could go either way: the parity fixtures are required to stay far from that. */
static double g_stop_margin = 1e300;
static void audit(double z1, double thr)
{
    if (thr > 0. && z1 > 0.) {
        const double m = fabs(z1 / thr - 1.);
        if (m < g_stop_margin) g_stop_margin = m;
    }
}
double sos_oracle_stop_margin(void) { return g_stop_margin; }

static double conv_term(double a, double d, double gg, double x3, double z1)
{
    if (a != 0.0 && d != 0.0 && x3 != 0.0) {
        double y = ((gg / d - d / a) / ((1 - gg / d) * (1 - gg / d)) * (gg / x3));
        z1 = fmax(z1, fabs(y));
    }
    return z1;
}

static double queue_term(double d, double gg) /* SOS_AJOUT_QUEUE, SOS_OS.F:3959-3975 */
{
    if (d == 0.) return 0.;
    return gg / (1 - gg / d);
}

int sos_oracle_os(int n, const double *mu, const double *ga_in, int os_nb, int nt,
                  int n0, double tetas, double ro, int imat_surf, int ifresnel, double ind_surf,
                  const double *h, const double *xdel, const double *ydel, const double *zprof, double ron,
                  const double *alpha_in, const double *beta_in, const double *gamma_in, const double *zeta_in,
                  double zout, int igmax, int iborm, int ipolar, const float *rsurf,
                  double *rec, int *n_orders, int *ig_last, double *emoins, double *eplus)
{
    geom_t gs, *g = &gs;
    const int W = 2 * n + 1, L = nt + 1;
    const size_t FS = (size_t)W * L;
    int i, j, k, is, ig, ier = 0;
    g_stop_margin = 1e300;
    g->n = n; g->nt = nt; g->L = L; g->W = W;
    g->rmu = calloc(W, sizeof(double));
    g->ga = calloc(W, sizeof(double));
    for (j = 1; j <= n; j++) {
        V(g->rmu, j) = mu[j - 1]; V(g->rmu, -j) = -mu[j - 1]; /* SOS_PREPA_OS.F:541-546 */
        V(g->ga, j) = ga_in[j - 1]; V(g->ga, -j) = ga_in[j - 1];
    }
    double *alpha = malloc(sizeof(double) * (os_nb + 1)), *beta = malloc(sizeof(double) * (os_nb + 1));
    double *gamma = malloc(sizeof(double) * (os_nb + 1)), *zeta = malloc(sizeof(double) * (os_nb + 1));
    memcpy(alpha, alpha_in, sizeof(double) * (os_nb + 1)); memcpy(beta, beta_in, sizeof(double) * (os_nb + 1));
    memcpy(gamma, gamma_in, sizeof(double) * (os_nb + 1)); memcpy(zeta, zeta_in, sizeof(double) * (os_nb + 1));

    /* fields */
    double *fld = calloc(FS * 24, sizeof(double));
    double *i1 = fld, *q1 = fld + FS, *u1 = fld + 2 * FS, *i2 = fld + 3 * FS, *q2 = fld + 4 * FS, *u2 = fld + 5 * FS;
    double *i1f = fld + 6 * FS, *q1f = fld + 7 * FS, *u1f = fld + 8 * FS;
    double *i3o = fld + 9 * FS, *q3o = fld + 10 * FS, *u3o = fld + 11 * FS;
    double *d1o = fld + 12 * FS, *e1o = fld + 13 * FS, *f1o = fld + 14 * FS;
    double *g1o = fld + 15 * FS, *h1o = fld + 16 * FS, *p1o = fld + 17 * FS;
    double *riio = fld + 18 * FS, *rqqo = fld + 19 * FS, *ruuo = fld + 20 * FS; /* (i,k) k=1..N used */
    double *vec = calloc((size_t)W * 32, sizeof(double));
    double *i3 = vec, *q3 = vec + W, *u3 = vec + 2 * W, *a1 = vec + 3 * W, *b1 = vec + 4 * W, *c1 = vec + 5 * W;
    double *d1 = vec + 6 * W, *e1 = vec + 7 * W, *f1 = vec + 8 * W, *g1 = vec + 9 * W, *h1 = vec + 10 * W, *p1 = vec + 11 * W;
    double *i4 = vec + 12 * W, *q4 = vec + 13 * W, *u4 = vec + 14 * W, *i5 = vec + 15 * W, *q5 = vec + 16 * W, *u5 = vec + 17 * W;
    double *xr = vec + 18 * W, *rii = vec + 19 * W, *rqq = vec + 20 * W, *ruu = vec + 21 * W;
    double *f11 = vec + 22 * W, *f12 = vec + 23 * W, *f33 = vec + 24 * W;
    double *i3z = vec + 25 * W, *q3z = vec + 26 * W, *u3z = vec + 27 * W;
    double *ch = calloc(L, sizeof(double));
    kern_t kn;
    double *kbuf = calloc((size_t)W * W * 6 + 3 * W, sizeof(double));
    kn.bp = kbuf; kn.gr = kbuf + (size_t)W * W; kn.gt = kbuf + (size_t)2 * W * W; kn.arr = kbuf + (size_t)3 * W * W;
    kn.art = kbuf + (size_t)4 * W * W; kn.att = kbuf + (size_t)5 * W * W;
    kn.xpl = kbuf + (size_t)6 * W * W; kn.xrl = kn.xpl + W; kn.xtl = kn.xrl + W;

    memset(rec, 0, sizeof(double) * (size_t)(iborm + 1) * 3 * W);
    for (is = 0; is <= iborm; is++) ig_last[is] = 0;
    *n_orders = 0; *emoins = 0.; *eplus = 0.;

    /* :678-699 */
    double aaa = ron / (2 - ron);
    aaa = (1 - aaa) / (1 + 2 * aaa);
    double beta0 = 1., beta2 = 0.5 * aaa, gamma2 = -aaa * sqrt(1.5), alpha2 = 3. * aaa;
    if (ipolar == 0) {
        gamma2 = 0.; alpha2 = 0.;
        for (k = 0; k <= os_nb; k++) { alpha[k] = 0.; gamma[k] = 0.; zeta[k] = 0.; }
    }
    /* :706-715 */
    double tab;
    if (n0 > 0) tab = -V(g->rmu, n0);
    else tab = -cos(acos(-1.0) * tetas / 180.);
    V(g->rmu, 0) = tab;
    if (tab == 0.0) goto done; /* :807 (limb incidence: returns with IER=0) */
    if (((zout < 0) && (zout != -1.0)) || (zout > TOA_ALT)) { ier = -1; goto done; } /* :811 */

    if (ifresnel == 1) { /* SOS_MAT_FRESNEL_PLAN_REFL :1753-1780 */
        for (j = 0; j <= n; j++) {
            double m = (j == 0) ? -V(g->rmu, 0) : V(g->rmu, j);
            double ind2 = ind_surf * ind_surf, mu2 = m * m;
            double x = sqrt(ind2 - 1.0 + mu2);
            double rl = (ind2 * m - x) / (ind2 * m + x);
            double rr = (m - x) / (m + x);
            f11[j] = (rl * rl + rr * rr) / 2.;
            if (ipolar == 1) { f12[j] = (rl * rl - rr * rr) / 2.; f33[j] = rl * rr; }
            else { f12[j] = 0.; f33[j] = 0.; }
        }
    }
    for (i = 0; i <= nt; i++) ch[i] = exp(-h[i] / (-tab)) / 4.; /* :837-839 */

    double sign = -1.;
    for (is = 0; is <= iborm; is++) { /* :872 */
        sign = -sign;
        if (is > 0) beta0 = 0.;
        for (j = -n; j <= n; j++) { V(i3, j) = 0.; V(q3, j) = 0.; V(u3, j) = 0.; }
        memset(i3o, 0, sizeof(double) * FS); memset(q3o, 0, sizeof(double) * FS); memset(u3o, 0, sizeof(double) * FS);
        /* surface matrices for this order :912-943 ; R_ab(I,J) = rs[ab][(J-1)*n + (I-1)] */
        const float *rs = NULL;
        float *rs_np = NULL;
        if (imat_surf == 1) {
            rs = rsurf + (size_t)is * 9 * n * n;
            if (ipolar == 0) {
                rs_np = malloc(sizeof(float) * 9 * n * n);
                memcpy(rs_np, rs, sizeof(float) * 9 * n * n);
                memset(rs_np + (size_t)n * n, 0, sizeof(float) * 8 * n * n);
                rs = rs_np;
            }
        }
#define RS(ab, I, J) rs[(size_t)(ab) * n * n + (size_t)((J)-1) * n + ((I)-1)]
        sos_oracle_noyaux(is, n, V(g->rmu, 0), mu, os_nb, alpha, beta, gamma, zeta,
                          kn.xpl, kn.xrl, kn.xtl, kn.bp, kn.gr, kn.gt, kn.arr, kn.art, kn.att);
        fsource_ordre1(g, is, xdel, ydel, beta0, beta2, gamma2, &kn, ch, i2, q2, u2);
        for (k = 1; k <= n; k++) { /* :970-992 */
            FLD(i1, nt, k) = 0.; FLD(q1, nt, k) = 0.; FLD(u1, nt, k) = 0.; V(xr, k) = 0.;
            if (!(ro == 0. || is != 0)) {
                FLD(i1, nt, k) = -ro * tab * exp(h[nt] / tab);
                V(xr, k) = FLD(i1, nt, k);
            }
            if (imat_surf == 1) {
                double rr = exp(h[nt] / tab) / V(g->rmu, k);
                FLD(i1, nt, k) = FLD(i1, nt, k) + RS(0, n0, k) * rr;
                FLD(q1, nt, k) = RS(3, n0, k) * rr;
                FLD(u1, nt, k) = RS(6, n0, k) * rr;
            }
        }
        integr_epopt(g, h, i2, q2, u2, i1, q1, u1);
        if (ifresnel == 1) { /* :1010-1043 */
            fsource_diff_fresnel1(g, is, f11[0], f12[0], xdel, ydel, beta0, beta2, gamma2, alpha2, &kn, tab, h, i2, q2, u2);
            for (k = 1; k <= n; k++) { FLD(i1f, nt, k) = 0.; FLD(q1f, nt, k) = 0.; FLD(u1f, nt, k) = 0.; }
            integr_epopt(g, h, i2, q2, u2, i1f, q1f, u1f);
            for (i = 0; i <= nt; i++)
                for (k = -n; k <= n; k++) {
                    FLD(i1, i, k) = FLD(i1, i, k) + FLD(i1f, i, k);
                    FLD(q1, i, k) = FLD(q1, i, k) + FLD(q1f, i, k);
                    FLD(u1, i, k) = FLD(u1, i, k) + FLD(u1f, i, k);
                }
        }
        for (k = 1; k <= n; k++) { /* :1051-1084 */
            V(rii, k) = 0.; V(rqq, k) = 0.; V(ruu, k) = 0.;
            for (i = 0; i <= nt; i++) { FLD(riio, i, k) = 0.; FLD(rqqo, i, k) = 0.; FLD(ruuo, i, k) = 0.; }
        }
        if (imat_surf == 1) {
            for (k = 1; k <= n; k++) {
                double a;
                for (i = 0; i <= nt; i++) {
                    a = -(h[nt] - h[i]) / V(g->rmu, k);
                    a = exp(a);
                    FLD(riio, i, k) = a * (FLD(i1, nt, k) - V(xr, k));
                    FLD(rqqo, i, k) = a * FLD(q1, nt, k);
                    FLD(ruuo, i, k) = a * FLD(u1, nt, k);
                }
                a = -h[nt] / V(g->rmu, k);
                a = exp(a);
                V(rii, k) = a * (FLD(i1, nt, k) - V(xr, k));
                V(rqq, k) = a * FLD(q1, nt, k);
                V(ruu, k) = a * FLD(u1, nt, k);
            }
        }
        for (k = -n; k <= -1; k++) { /* :1094-1113 */
            V(i3, k) = FLD(i1, nt, k); V(q3, k) = FLD(q1, nt, k); V(u3, k) = FLD(u1, nt, k);
            V(d1, k) = FLD(i1, nt, k); V(e1, k) = FLD(q1, nt, k); V(f1, k) = FLD(u1, nt, k);
            for (i = 0; i <= nt; i++) { FLD(i3o, i, k) = FLD(i1, i, k); FLD(q3o, i, k) = FLD(q1, i, k); FLD(u3o, i, k) = FLD(u1, i, k); }
        }
        for (k = 1; k <= n; k++) { /* :1118-1137 */
            V(i3, k) = FLD(i1, 0, k); V(q3, k) = FLD(q1, 0, k); V(u3, k) = FLD(u1, 0, k);
            V(d1, k) = FLD(i1, 0, k); V(e1, k) = FLD(q1, 0, k); V(f1, k) = FLD(u1, 0, k);
            for (i = 0; i <= nt; i++) { FLD(i3o, i, k) = FLD(i1, i, k); FLD(q3o, i, k) = FLD(q1, i, k); FLD(u3o, i, k) = FLD(u1, i, k); }
        }
        ig = 1;
        ig_last[is] = 1;
        for (;;) { /* label 503 */
            double z1, lsol;
            ig = ig + 1;
            if (ig > igmax) break;
            ig_last[is] = ig;
            fsource_ordreig(g, is, xdel, ydel, beta0, beta2, gamma2, alpha2, &kn, i1, q1, u1, i2, q2, u2);
            for (k = 1; k <= n; k++) { FLD(i1, nt, k) = 0.; FLD(q1, nt, k) = 0.; FLD(u1, nt, k) = 0.; V(xr, k) = 0.; }
            lsol = 0.;
            for (j = 1; j <= n; j++) lsol = lsol + V(g->ga, j) * FLD(i1, nt, -j) * V(g->rmu, j);
            lsol = 2 * lsol * ro;
            if (!(ro == 0. || is != 0))
                for (j = 1; j <= n; j++) { FLD(i1, nt, j) = lsol; V(xr, j) = lsol; }
            if (imat_surf == 1) { /* :1194-1220 */
                for (k = 1; k <= n; k++) {
                    double ii2 = 0., qq2 = 0., uu2 = 0., rrmu = 2 / V(g->rmu, k);
                    for (j = 1; j <= n; j++) {
                        double z = V(g->ga, j);
                        double xi1 = FLD(i1, nt, -j), xq1 = FLD(q1, nt, -j), xu1 = FLD(u1, nt, -j);
                        ii2 = ii2 + z * (xi1 * RS(0, j, k) + xq1 * RS(1, j, k) + xu1 * RS(2, j, k));
                        qq2 = qq2 + z * (xi1 * RS(3, j, k) + xq1 * RS(4, j, k) + xu1 * RS(5, j, k));
                        uu2 = uu2 + z * (xi1 * RS(6, j, k) + xq1 * RS(7, j, k) + xu1 * RS(8, j, k));
                    }
                    FLD(i1, nt, k) = ii2 * rrmu + V(xr, k);
                    FLD(q1, nt, k) = qq2 * rrmu;
                    FLD(u1, nt, k) = uu2 * rrmu;
                }
            }
            if (ifresnel == 1) { /* :1225-1239 */
                for (k = 1; k <= n; k++) {
                    FLD(i1, nt, k) = FLD(i1, nt, k) + f11[k] * FLD(i1, nt, -k) + f12[k] * FLD(q1, nt, -k);
                    FLD(q1, nt, k) = FLD(q1, nt, k) + f12[k] * FLD(i1, nt, -k) + f11[k] * FLD(q1, nt, -k);
                    FLD(u1, nt, k) = FLD(u1, nt, k) + f33[k] * FLD(u1, nt, -k);
                }
            }
            integr_epopt(g, h, i2, q2, u2, i1, q1, u1);
            for (k = -n; k <= -1; k++) { /* :1248-1263 */
                V(g1, k) = FLD(i1, nt, k); V(h1, k) = FLD(q1, nt, k); V(p1, k) = FLD(u1, nt, k);
                for (i = 0; i <= nt; i++) { FLD(g1o, i, k) = FLD(i1, i, k); FLD(h1o, i, k) = FLD(q1, i, k); FLD(p1o, i, k) = FLD(u1, i, k); }
            }
            for (k = 1; k <= n; k++) {
                V(g1, k) = FLD(i1, 0, k); V(h1, k) = FLD(q1, 0, k); V(p1, k) = FLD(u1, 0, k);
                for (i = 0; i <= nt; i++) { FLD(g1o, i, k) = FLD(i1, i, k); FLD(h1o, i, k) = FLD(q1, i, k); FLD(p1o, i, k) = FLD(u1, i, k); }
            }
            if (ig != 2) {
                z1 = 0.; /* SOS_PARAM_CONV :3428-3458 */
                for (k = -n; k <= n; k++) {
                    if (k == 0) continue;
                    z1 = conv_term(V(a1, k), V(d1, k), V(g1, k), V(i3, k), z1);
                    z1 = conv_term(V(b1, k), V(e1, k), V(h1, k), V(q3, k), z1);
                    z1 = conv_term(V(c1, k), V(f1, k), V(p1, k), V(u3, k), z1);
                }
                audit(z1, SEUIL_CV_SG);
                if (!(z1 > SEUIL_CV_SG)) { /* SOS_AJOUT_QUEUE :3955-4015 */
                    for (j = -n; j <= n; j++) {
                        if (j == 0) continue;
                        V(i3, j) = V(i3, j) + queue_term(V(d1, j), V(g1, j));
                        V(q3, j) = V(q3, j) + queue_term(V(e1, j), V(h1, j));
                        V(u3, j) = V(u3, j) + queue_term(V(f1, j), V(p1, j));
                    }
                    for (j = -n; j <= n; j++) {
                        if (j == 0) continue;
                        for (i = 0; i <= nt; i++) {
                            FLD(i3o, i, j) = FLD(i3o, i, j) + queue_term(FLD(d1o, i, j), FLD(g1o, i, j));
                            FLD(q3o, i, j) = FLD(q3o, i, j) + queue_term(FLD(e1o, i, j), FLD(h1o, i, j));
                            FLD(u3o, i, j) = FLD(u3o, i, j) + queue_term(FLD(f1o, i, j), FLD(p1o, i, j));
                        }
                    }
                    break; /* GO TO 505 */
                }
            }
            /* label 506 :1323-1363 */
            for (k = -n; k <= n; k++) {
                V(a1, k) = V(d1, k); V(b1, k) = V(e1, k); V(c1, k) = V(f1, k);
                V(d1, k) = V(g1, k); V(e1, k) = V(h1, k); V(f1, k) = V(p1, k);
                for (i = 0; i <= nt; i++) { FLD(d1o, i, k) = FLD(g1o, i, k); FLD(e1o, i, k) = FLD(h1o, i, k); FLD(f1o, i, k) = FLD(p1o, i, k); }
            }
            for (j = 1; j <= n; j++) {
                V(i3, j) = V(i3, j) + FLD(i1, 0, j); V(q3, j) = V(q3, j) + FLD(q1, 0, j); V(u3, j) = V(u3, j) + FLD(u1, 0, j);
                V(i3, -j) = V(i3, -j) + FLD(i1, nt, -j); V(q3, -j) = V(q3, -j) + FLD(q1, nt, -j); V(u3, -j) = V(u3, -j) + FLD(u1, nt, -j);
                for (i = 0; i <= nt; i++) {
                    FLD(i3o, i, j) += FLD(i1, i, j); FLD(q3o, i, j) += FLD(q1, i, j); FLD(u3o, i, j) += FLD(u1, i, j);
                    FLD(i3o, i, -j) += FLD(i1, i, -j); FLD(q3o, i, -j) += FLD(q1, i, -j); FLD(u3o, i, -j) += FLD(u1, i, -j);
                }
            }
            z1 = 0.; /* SOS_ARRET_DIFFUS_1 :3524-3543 */
            for (k = -n; k <= n; k++) {
                if (k == 0) continue;
                int ind = (k < 0) ? nt : 0;
                z1 = fmax(z1, fabs(FLD(i1, ind, k)));
                z1 = fmax(z1, fabs(FLD(q1, ind, k)));
                z1 = fmax(z1, fabs(FLD(u1, ind, k)));
            }
            audit(z1, SEUIL_VALDIF);
            if (!(z1 > SEUIL_VALDIF)) break;
            z1 = 0.; /* SOS_ARRET_DIFFUS_2 :3626-3656 */
            for (k = -n; k <= n; k++) {
                if (k == 0) continue;
                int ind = (k < 0) ? nt : 0;
                if (V(i3, k) != 0.0) z1 = fmax(z1, fabs(FLD(i1, ind, k) / V(i3, k)));
                if (V(q3, k) != 0.0) z1 = fmax(z1, fabs(FLD(q1, ind, k) / V(q3, k)));
                if (V(u3, k) != 0.0) z1 = fmax(z1, fabs(FLD(u1, ind, k) / V(u3, k)));
            }
            audit(z1, SEUIL_SUMDIF);
            if (!(z1 > SEUIL_SUMDIF)) break;
            if (!(ig < igmax)) break; /* :1406 */
        }
        /* label 505 */
        if (imat_surf == 1) { /* :1421-1439 */
            for (j = 1; j <= n; j++) {
                V(i3, j) -= V(rii, j); V(q3, j) -= V(rqq, j); V(u3, j) -= V(ruu, j);
                for (i = 0; i <= nt; i++) { FLD(i3o, i, j) -= FLD(riio, i, j); FLD(q3o, i, j) -= FLD(rqqo, i, j); FLD(u3o, i, j) -= FLD(ruuo, i, j); }
            }
        }
        if (is == 0) { /* :1447-1456 */
            double em = 0., ep = 0.;
            for (j = 1; j <= n; j++) {
                em = em + V(g->rmu, j) * V(g->ga, j) * V(i3, -j);
                ep = ep + V(g->rmu, j) * V(g->ga, j) * V(i3, j);
            }
            *emoins = -em * 2 / tab;
            *eplus = -ep * 2 / tab;
        }
        {
            double coef = 2.;
            if (is == 0) coef = 1.;
            for (j = -n; j <= n; j++) { /* :1463-1473 */
                if (j == 0) continue;
                V(i4, j) = V(i4, j) + coef * V(i3, j); V(q4, j) = V(q4, j) + coef * V(q3, j); V(u4, j) = V(u4, j) + coef * V(u3, j);
                V(i5, j) = V(i5, j) + coef * V(i3, j) * sign; V(q5, j) = V(q5, j) + coef * V(q3, j) * sign; V(u5, j) = V(u5, j) + coef * V(u3, j) * sign;
            }
        }
        if (zout == -1) { /* :1484-1534 */
            for (k = -n; k <= -1; k++) { V(i3z, k) = FLD(i3o, nt, k); V(q3z, k) = FLD(q3o, nt, k); V(u3z, k) = FLD(u3o, nt, k); }
            for (k = 1; k <= n; k++) { V(i3z, k) = FLD(i3o, 0, k); V(q3z, k) = FLD(q3o, 0, k); V(u3z, k) = FLD(u3o, 0, k); }
        } else {
            double zz;
            j = 1;
            while (zout < zprof[j]) j = j + 1;
            zz = (zout - zprof[j - 1]) / (zprof[j] - zprof[j - 1]);
            for (k = -n; k <= n; k++) {
                V(i3z, k) = (1 - zz) * FLD(i3o, j - 1, k) + zz * FLD(i3o, j, k);
                V(q3z, k) = (1 - zz) * FLD(q3o, j - 1, k) + zz * FLD(q3o, j, k);
                V(u3z, k) = (1 - zz) * FLD(u3o, j - 1, k) + zz * FLD(u3o, j, k);
            }
        }
        {
            double *r = rec + (size_t)is * 3 * W;
            for (k = -n; k <= n; k++) {
                if (k == 0) continue;
                r[0 * W + k + n] = V(i3z, k); r[1 * W + k + n] = V(q3z, k); r[2 * W + k + n] = V(u3z, k);
            }
        }
        *n_orders = is + 1;
        free(rs_np);
        {
            double z1 = 0.; /* SOS_ARRET_FOURIER :3759-3793 */
            for (j = -n; j <= n; j++) {
                if (j == 0) continue;
                if (V(q4, j) != 0.0) z1 = fmax(z1, fabs(V(q3, j) / V(q4, j)));
                if (V(i4, j) != 0.0) z1 = fmax(z1, fabs(V(i3, j) / V(i4, j)));
                if (V(u4, j) != 0.0) z1 = fmax(z1, fabs(V(u3, j) / V(u4, j)));
                if (V(q5, j) != 0.0) z1 = fmax(z1, fabs(V(q3, j) / V(q5, j)));
                if (V(u5, j) != 0.0) z1 = fmax(z1, fabs(V(u3, j) / V(u5, j)));
                if (V(i5, j) != 0.0) z1 = fmax(z1, fabs(V(i3, j) / V(i5, j)));
            }
            audit(z1, SEUIL_SF);
            if (!(z1 > SEUIL_SF)) break;
        }
    }
#undef RS
done:
    free(g->rmu); free(g->ga); free(alpha); free(beta); free(gamma); free(zeta);
    free(fld); free(vec); free(ch); free(kbuf);
    return ier;
}

/* SOS.F:523-550 */
int sos_oracle_profile_rescale(int nt, double a_tronc, double piz, double piztr, int os_nb,
                               double *h, double *xdel, double *ydel)
{
    int i, lta = 1;
    double *htr = malloc(sizeof(double) * (nt + 1));
    htr[0] = h[0];
    if (a_tronc != 0.) {
        for (i = 1; i <= nt; i++) {
            double va = xdel[i] * (h[i] - h[i - 1]);
            double vatr = va * (1 - piz * 0.5 * a_tronc);
            double vr = ydel[i] * (h[i] - h[i - 1]);
            double vg = (1 - xdel[i] - ydel[i]) * (h[i] - h[i - 1]);
            htr[i] = (vatr + vr + vg) + htr[i - 1];
            xdel[i] = vatr / (vatr + vr + vg);
            ydel[i] = vr / (vatr + vr + vg);
        }
    }
    for (i = 0; i <= nt; i++) {
        if (a_tronc != 0.) h[i] = htr[i];
        xdel[i] = xdel[i] * piztr;
        if (xdel[i] != 0.) lta = 0;
    }
    free(htr);
    return lta ? 2 : os_nb;
}

/* SOS_AGGREGATE.F:372-488, applied serially over the bin list (SOS_PROC.F:3564). */
int sos_oracle_aggregate(int nb, int fmax, int w, const int *nf, const double *aik,
                         const double *rec_bins, const double *scal_bins,
                         double *out_rec, double *out_scal)
{
    int b, s, c, nfo = 0;
    size_t rs = (size_t)3 * w;
    memset(out_rec, 0, sizeof(double) * fmax * rs);
    for (c = 0; c < 7; c++) out_scal[c] = 0.;
    for (b = 0; b < nb; b++) {
        const double *rb = rec_bins + (size_t)b * fmax * rs;
        const double *sb = scal_bins + (size_t)b * 7;
        int nfb = nf[b];
        int nmax = nfb > nfo ? nfb : nfo;
        for (s = 0; s < nmax; s++) {
            size_t e;
            for (e = 0; e < rs; e++) {
                double t = (s < nfb) ? rb[s * rs + e] : 0.;
                double r = (s < nfo) ? out_rec[s * rs + e] : 0.;
                out_rec[s * rs + e] = r + aik[b] * t;
            }
        }
        nfo = nmax;
        out_scal[0] = out_scal[0] + aik[b] * sb[0]; /* TDIFMUS */
        out_scal[1] = out_scal[1] + aik[b] * sb[1]; /* EMOINS */
        out_scal[2] = out_scal[2] + aik[b] * sb[2]; /* EPLUS */
        for (c = 3; c < 6; c++) { /* TTOT_TRONC, TTOT_VRAI, TAUOUT :467-488 */
            double trans;
            if (out_scal[c] != 0) trans = aik[b] * exp(-sb[c]) + exp(-out_scal[c]);
            else trans = aik[b] * exp(-sb[c]);
            out_scal[c] = -log(trans);
        }
    }
    return nfo;
}
