/* oracle/sos_glitter_oracle.c -- CPU restatement of the Cox-Munk rough-sea reflection matrices.
 *
 * TEST INFRASTRUCTURE ONLY (see sos_oracle.h).
 *
 * Follows SOS_GLITTER (src/SOS_GLITTER.F:229-371): SOS_GSF (:451-711) + SOS_CALCG (:755-784),
 * SOS_MAT_FRESNEL (src/SOS_SURFACE.F:1235-1603), SOS_MAT_REFLEXION (:1708-1973) with
 * SOS_NOYAUX_FRESNEL (:2029-2227), SOS_MISE_FORMAT (:2307-2443), including the REAL*4 sub-expressions,
 * the 4(E15.8) text round trip of the Fresnel expansion coefficients and the REAL*4 storage of the
 * matrices.  No files: results are returned in memory in the layout of the GLITTER file records.
 */
#include "sos_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PH_NU 1024   /* SOS.h:319 */
#define PH_NQ 10     /* SOS.h:325 */
#define PH_TEST 10000 /* SOS.h:312 */

/* value after a Fortran E15.8 write + read (8 significant digits, round to nearest) */
static double e15_8(double x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.7E", x);
    return strtod(buf, NULL);
}

double sos_oracle_sigma2(double wind) /* SIG = .003 + .00512*WIND with REAL*4 literals (SOS_GLITTER.F:300) */
{
    return (double)0.003f + (double)0.00512f * wind;
}

static double calcg(double cs12, double c12, double s12, double sig, double phi) /* SOS_CALCG :779-781 */
{
    double costetad = -c12 + s12 * cos(phi);
    double x = (1 - costetad) / cs12;
    return x * x * exp(-(x - 1) / sig);
}

/* SOS_GSF for one pair (mu1 = RMU(I1), mu2 = RMU(I2)).  e[0..os_nm]; returns IL. */
int sos_oracle_gsf_pair(double mu1, double mu2, double sig, int os_nm, double *e)
{
    const double pi = acos(-1.0);
    static double u[PH_NU + 1];
    double c1 = mu1, s1 = sqrt(1 - c1 * c1), c2 = mu2, s2 = sqrt(1 - c2 * c2);
    double c12 = c1 * c2, s12 = s1 * s2, cs12 = (c1 + c2);
    double gmax, gmin, phib, q, t1 = 0., z, y, x;
    int i, is, il;
    cs12 = .5 * cs12 * cs12;
    u[0] = gmax = calcg(cs12, c12, s12, sig, 0.0);
    u[PH_NU] = gmin = calcg(cs12, c12, s12, sig, pi);
    x = PH_TEST * gmin;
    if (x >= gmax) { /* :568-578 */
        phib = pi;
        q = pi / PH_NU;
        for (i = 1; i <= PH_NU; i++) u[i] = calcg(cs12, c12, s12, sig, q * i);
    } else { /* bisection :586-638 */
        double phi1 = 0, phi2 = pi, g;
        for (;;) {
            phib = .5 * (phi1 + phi2);
            g = calcg(cs12, c12, s12, sig, phib);
            x = PH_TEST * g;
            if (fabs(x - gmax) < (double).01f * gmax) break;
            if (x <= gmax) phi2 = phib; else phi1 = phib;
        }
        q = phib / PH_NU;
        for (i = 1; i <= PH_NU; i++) u[i] = calcg(cs12, c12, s12, sig, q * i);
        gmin = u[PH_NU];
    }
    il = os_nm;
    for (is = 0; is <= os_nm; is++) { /* :644-678 */
        int ia = 1, j, k, ip;
        z = .5 * (gmax + gmin * cos(is * phib));
        for (i = 1; i <= PH_NQ; i++) {
            double xt;
            ia = 2 * ia;
            ip = PH_NU / ia;
            y = 0;
            for (j = 1; j <= ia; j += 2) {
                k = ip * j;
                y = y + u[k] * cos((is * k) * q);
            }
            y = 2 * y / ia;
            xt = fabs(z - y) / z;
            if (xt < (double).0001f) break;
            z = .5 * (y + z);
        }
        e[is] = phib * z / pi;
        if (is == 0) { t1 = e[0]; continue; }
        t1 = t1 + 2 * e[is];
        if (!(fabs(t1 - gmax) / gmax > (double).001f)) { il = is; break; }
    }
    return il;
}

/* SOS_MAT_FRESNEL: alpha,beta,gamma,zeta[0..os_ns] after the 4(E15.8) text round trip.
 * mu[0..n-1] = RMU(1..N), chr = Gauss weights (CHR(-j)=CHR(j)). */
void sos_oracle_mat_fresnel(int n, const double *mu, const double *chr, double ind, int os_ns,
                            double *alpha, double *beta, double *gamma, double *zeta)
{
    int j, k, i;
    double *delta = calloc(os_ns + 3, sizeof(double));
    double *pl = calloc(os_ns + 4, sizeof(double)) /* PL(-1:..) at +1 */, *pol = calloc(os_ns + 3, sizeof(double));
    double *r11 = calloc(2 * n + 1, sizeof(double)), *r12 = calloc(2 * n + 1, sizeof(double)), *r33 = calloc(2 * n + 1, sizeof(double));
#define PL(k) pl[(k) + 1]
#define RMUJ(j) ((j) > 0 ? mu[(j)-1] : -mu[-(j)-1])
#define CHRJ(j) ((j) > 0 ? chr[(j)-1] : chr[-(j)-1])
    for (k = 0; k <= os_ns; k++) { beta[k] = gamma[k] = alpha[k] = zeta[k] = 0.; }
    for (j = -n; j <= n; j++) { /* :1346-1381 */
        double c, a, b, rl, rr;
        if (j == 0) continue;
        c = RMUJ(j);
        c = sqrt(.5 * (1 + c));
        a = sqrt(ind * ind - 1.0 + c * c);
        b = ind * ind * c;
        rl = -(b - a) / (b + a);
        rr = (c - a) / (c + a);
        r11[j + n] = .5 * (rl * rl + rr * rr);
        r12[j + n] = .5 * (rl * rl - rr * rr);
        r33[j + n] = rl * rr;
    }
    for (j = -n; j <= n; j++) { /* :1387-1400 */
        double x, xrmu;
        if (j == 0) continue;
        x = r11[j + n] * CHRJ(j);
        xrmu = RMUJ(j);
        PL(-1) = 0.; PL(0) = 1.;
        for (k = 0; k <= os_ns; k++) {
            PL(k + 1) = ((2 * k + 1.) * xrmu * PL(k) - k * PL(k - 1)) / (k + 1.);
            beta[k] = beta[k] + x * PL(k);
        }
    }
    for (k = 0; k <= os_ns; k++) beta[k] = (2 * k + 1) * beta[k] * .5;
    for (j = -n; j <= n; j++) { /* :1433-1456 */
        double xxx, xx, xrmu;
        if (j == 0) continue;
        xxx = CHRJ(j) * r12[j + n];
        xx = CHRJ(j) * r33[j + n];
        pol[0] = 0.; pol[1] = 0.;
        xrmu = RMUJ(j);
        PL(-1) = 0.; PL(0) = 1.;
        pol[2] = 3. * (1. - xrmu * xrmu) / 2. / sqrt(6.0);
        for (k = 2; k <= os_ns; k++) {
            double d = (2. * k + 1.) / sqrt(1.0 * (k + 3.) * (k - 1.));
            double e = sqrt(1.0 * (k + 2.) * (k - 2.)) / (2. * k + 1.);
            pol[k + 1] = d * (xrmu * pol[k] - e * pol[k - 1]);
            gamma[k] = gamma[k] + xxx * pol[k];
        }
        for (k = 0; k <= os_ns; k++) {
            PL(k + 1) = ((2. * k + 1.) * xrmu * PL(k) - k * PL(k - 1)) / (k + 1.);
            delta[k] = delta[k] + xx * PL(k);
        }
    }
    for (k = 0; k <= os_ns; k++) {
        delta[k] = delta[k] * (2. * k + 1.) * .5;
        gamma[k] = gamma[k] * (2. * k + 1.) * .5;
    }
    for (i = 2; i <= os_ns; i++) { /* :1521-1546 */
        /* REAL*4: CO1 and CO2 have only REAL*4/INTEGER operands (:1522-1523) */
        float co1f = 4 * (2 * i + 1.f) / (float)i / (i - 1.f) / (i + 1.f) / (i + 2.f);
        float co2f = i * (i - 1.f) / ((i + 1.f) * (i + 2.f));
        double co1 = co1f, co2 = co2f, co3;
        int nn = (int)(i * .5f), mm = (int)((i - 1) * .5f);
        double som1 = 0., som2 = 0., som3 = 0., som4 = 0.;
        co3 = co2 * delta[i];
        co2 = co2 * beta[i];
        for (j = 1; j <= nn; j++) {
            double x2 = (double)((i - 1.f) * (i - 1.f) - 3.f * (2 * j - 1.f) * (i - j));
            som1 = som1 + x2 * beta[i - 2 * j];
            som2 = som2 + x2 * delta[i - 2 * j];
        }
        for (j = 0; j <= mm; j++) {
            double x2 = (double)((i - 1.f) * (i - 1.f) - 3.f * j * (2 * i - 2 * j - 1.f));
            som3 = som3 + x2 * beta[i - 2 * j - 1];
            som4 = som4 + x2 * delta[i - 2 * j - 1];
        }
        zeta[i] = co3 - co1 * (som2 - som3);
        alpha[i] = co2 - co1 * (som1 - som4);
    }
    for (k = 0; k <= os_ns; k++) { /* WRITE 4(E15.8) :1552 / READ :1822 */
        alpha[k] = e15_8(alpha[k]); beta[k] = e15_8(beta[k]); gamma[k] = e15_8(gamma[k]); zeta[k] = e15_8(zeta[k]);
    }
#undef PL
#undef RMUJ
#undef CHRJ
    free(delta); free(pl); free(pol); free(r11); free(r12); free(r33);
}

/* SOS_NOYAUX_FRESNEL (:2029-2227).  Outputs X[(is)*2 + (k-1)], is = 0..os_ns, k = 1,2. */
static void noyaux_fresnel(double rmu1, double rmu2, int os_ns, const double *alpha, const double *beta,
                           const double *gamma, const double *zeta,
                           double *bp, double *gr, double *gt, double *arr, double *art, double *att)
{
    const int NL = os_ns + 2;
    double *psl = calloc((size_t)NL * 2, sizeof(double)), *rsl = calloc((size_t)NL * 2, sizeof(double)), *tsl = calloc((size_t)NL * 2, sizeof(double));
#define P(l, j) psl[((j)-1) * NL + (l) + 1]
#define R(l, j) rsl[((j)-1) * NL + (l) + 1]
#define T(l, j) tsl[((j)-1) * NL + (l) + 1]
    const double rac3 = sqrt(3.0), x26 = 2. * sqrt(6.0);
    double r[3];
    int is, j, k, l, i;
    r[1] = rmu1; r[2] = rmu2;
    for (is = 0; is <= os_ns; is++) {
        if (is == 0) {
            for (j = 1; j <= 2; j++) {
                double c = r[j], x;
                P(0, j) = 1; P(1, j) = c;
                x = (3 * c * c - 1) * 0.5; P(2, j) = x;
                R(1, j) = 0;
                x = 3 * (1 - c * c) / x26; R(2, j) = x;
                T(1, j) = 0.; T(2, j) = 0.;
            }
        } else if (is == 1) {
            for (j = 1; j <= 2; j++) {
                double c = r[j], x = 1 - c * c;
                P(0, j) = 0; P(1, j) = sqrt(x * 0.5); P(2, j) = c * P(1, j) * rac3;
                T(1, j) = 0.; R(1, j) = 0;
                R(2, j) = -c * sqrt(x) * 0.5;
                T(2, j) = -sqrt(x) * 0.5;
            }
        } else {
            double a = 1, b;
            for (i = 1; i <= is; i++) { double x = i; a = a * sqrt((i + is) / x) * 0.5; }
            b = a * sqrt(is / (is + 1.0)) * sqrt((is - 1.0) / (is + 2.));
            for (j = 1; j <= 2; j++) {
                double c = r[j], xx = 1 - c * c, yy = is * 0.5, x;
                P(is - 1, j) = 0.; R(is - 1, j) = 0.; T(is - 1, j) = 0.;
                x = pow(xx, yy);
                P(is, j) = a * x;
                yy = yy - 1;
                x = pow(xx, yy);
                R(is, j) = b * (1 + c * c) * x;
                T(is, j) = 2 * b * c * x;
            }
        }
        k = 2;
        if (is > 2) k = is;
        for (l = k; l <= os_ns - 1; l++) {
            double a = (2 * l + 1.) / sqrt((l + is + 1.0) * (l - is + 1.));
            double b = sqrt((double)((l + is) * (l - is))) / (2. * l + 1.);
            double d = (l + 1.) * (2 * l + 1.) / sqrt((l + 3.0) * (l - 1.) * (l + is + 1.) * (l - is + 1.));
            double e = sqrt((l + 2.0) * (l - 2.) * (l + is) * (l - is)) / (l * (2. * l + 1.));
            double f = (double)((2.f * (float)is) / ((float)l * ((float)l + 1.f))); /* REAL*4 :2176 */
            for (j = 1; j <= 2; j++) {
                double c = r[j], x;
                x = a * (c * P(l, j) - b * P(l - 1, j)); P(l + 1, j) = x;
                x = d * (c * R(l, j) - f * T(l, j) - e * R(l - 1, j)); R(l + 1, j) = x;
                x = d * (c * T(l, j) - f * R(l, j) - e * T(l - 1, j)); T(l + 1, j) = x;
            }
        }
        for (k = 1; k <= 2; k++) {
            double sbp = 0., sarr = 0., satt = 0., sgr = 0., sgt = 0., sart = 0.;
            j = 3 - k;
            for (l = is; l <= os_ns; l++) {
                sbp = sbp + beta[l] * P(l, j) * P(l, k);
                sgr = sgr + gamma[l] * P(l, j) * R(l, k);
                sgt = sgt + gamma[l] * P(l, j) * T(l, k);
                satt = satt + alpha[l] * T(l, j) * T(l, k) + zeta[l] * R(l, j) * R(l, k);
                sarr = sarr + zeta[l] * T(l, j) * T(l, k) + alpha[l] * R(l, j) * R(l, k);
                sart = sart + alpha[l] * R(l, k) * T(l, j) + zeta[l] * R(l, j) * T(l, k);
            }
            bp[is * 2 + k - 1] = sbp; arr[is * 2 + k - 1] = sarr; att[is * 2 + k - 1] = satt;
            gr[is * 2 + k - 1] = sgr; gt[is * 2 + k - 1] = sgt; art[is * 2 + k - 1] = sart;
        }
    }
#undef P
#undef R
#undef T
    free(psl); free(rsl); free(tsl);
}

/* SOS_GLITTER end to end.  out: REAL*4 [os_nb+1][9][N][N] in GLITTER-file record order
 * out[s][ab][(J-1)*N + (I-1)] = P_ab(I,J) (SOS_SURFACE.F:2404-2412).
 * il_out (optional): IL per pair in (I1, I2<=I1) order; e_out (optional): [npairs][os_nm+1]. */
int sos_oracle_glitter(int n, const double *mu, const double *chr, double wind, double ind,
                       int os_nb, int os_ns, int os_nm, float *out, int *il_out, double *e_out,
                       double *coef_out /* [4][os_ns+1] alpha,beta,gamma,zeta or NULL */)
{
    const double sig = sos_oracle_sigma2(wind);
    const double coef = (1. / sig);
    double *alpha = calloc(os_ns + 1, sizeof(double)), *beta = calloc(os_ns + 1, sizeof(double));
    double *gamma = calloc(os_ns + 1, sizeof(double)), *zeta = calloc(os_ns + 1, sizeof(double));
    double *g = calloc(os_nm + os_ns + os_nb + 2, sizeof(double));
    double *kb = calloc((size_t)12 * (os_ns + 1), sizeof(double));
    double *bp = kb, *gr = kb + 2 * (os_ns + 1), *gt = kb + 4 * (os_ns + 1), *arr = kb + 6 * (os_ns + 1);
    double *art = kb + 8 * (os_ns + 1), *att = kb + 10 * (os_ns + 1);
    int i, j, is, k, pair = 0;
    sos_oracle_mat_fresnel(n, mu, chr, ind, os_ns, alpha, beta, gamma, zeta);
    if (coef_out) {
        memcpy(coef_out, alpha, sizeof(double) * (os_ns + 1));
        memcpy(coef_out + (os_ns + 1), beta, sizeof(double) * (os_ns + 1));
        memcpy(coef_out + 2 * (os_ns + 1), gamma, sizeof(double) * (os_ns + 1));
        memcpy(coef_out + 3 * (os_ns + 1), zeta, sizeof(double) * (os_ns + 1));
    }
#define BPk(K, c) bp[(K)*2 + (c)-1]
#define GRk(K, c) gr[(K)*2 + (c)-1]
#define GTk(K, c) gt[(K)*2 + (c)-1]
#define ARRk(K, c) arr[(K)*2 + (c)-1]
#define ARTk(K, c) art[(K)*2 + (c)-1]
#define ATTk(K, c) att[(K)*2 + (c)-1]
#define OUT(s, ab, I, J) out[(((size_t)(s)*9 + (ab)) * n + ((J)-1)) * n + ((I)-1)]
    for (i = 1; i <= n; i++) {
        for (j = 1; j <= i; j++, pair++) {
            int lim = sos_oracle_gsf_pair(mu[i - 1], mu[j - 1], sig, os_nm, g);
            if (il_out) il_out[pair] = lim;
            if (e_out) { memset(e_out + (size_t)pair * (os_nm + 1), 0, sizeof(double) * (os_nm + 1)); memcpy(e_out + (size_t)pair * (os_nm + 1), g, sizeof(double) * (lim + 1)); }
            for (k = lim + 1; k <= os_nm; k++) g[k] = 0.;
            noyaux_fresnel(mu[i - 1], mu[j - 1], os_ns, alpha, beta, gamma, zeta, bp, gr, gt, arr, art, att);
            for (is = 0; is <= os_nb; is++) { /* SOS_MAT_REFLEXION :1864-1933 */
                double x = coef * g[is] / 4., y;
                double r111 = x * BPk(0, 1), r121 = x * GRk(0, 1), r122 = x * GRk(0, 2), r131 = 0., r132 = 0., r231 = 0., r232 = 0.;
                double r211 = x * GRk(0, 2), r212 = x * GRk(0, 1), r221 = x * ARRk(0, 2), r222 = x * ARRk(0, 1);
                double r311 = 0., r312 = 0., r321 = 0., r322 = 0., r331 = x * ATTk(0, 2), r332 = x * ATTk(0, 1);
                int im = 1;
                for (k = 1; k <= os_ns; k++) {
                    int i1 = k + is, i2 = abs(k - is);
                    im = -im;
                    if ((i1 > lim) && (i2 > lim)) continue;
                    x = coef * im * (g[i1] + g[i2]) / 4.;
                    y = coef * im * (g[i2] - g[i1]) / 4.;
                    r111 = r111 + BPk(k, 1) * x;
                    r121 = r121 + GRk(k, 1) * x; r122 = r122 + GRk(k, 2) * x;
                    r131 = r131 + GTk(k, 1) * y; r132 = r132 + GTk(k, 2) * y;
                    r211 = r211 + GRk(k, 2) * x; r212 = r212 + GRk(k, 1) * x;
                    r221 = r221 + ARRk(k, 2) * x; r222 = r222 + ARRk(k, 1) * x;
                    r231 = r231 + ARTk(k, 2) * y; r232 = r232 + ARTk(k, 1) * y;
                    r311 = r311 + GTk(k, 2) * y; r312 = r312 + GTk(k, 1) * y;
                    r321 = r321 + ARTk(k, 1) * y; r322 = r322 + ARTk(k, 2) * y;
                    r331 = r331 + ATTk(k, 2) * x; r332 = r332 + ATTk(k, 1) * x;
                }
                /* M(IS,1) -> P(I,J), M(IS,2) -> P(J,I) (SOS_MISE_FORMAT :2378-2395; for I = J the second wins) */
                OUT(is, 0, i, j) = (float)r111; OUT(is, 0, j, i) = (float)r111;
                OUT(is, 1, i, j) = (float)r121; OUT(is, 1, j, i) = (float)r122;
                OUT(is, 2, i, j) = (float)r131; OUT(is, 2, j, i) = (float)r132;
                OUT(is, 3, i, j) = (float)r211; OUT(is, 3, j, i) = (float)r212;
                OUT(is, 4, i, j) = (float)r221; OUT(is, 4, j, i) = (float)r222;
                OUT(is, 5, i, j) = (float)r231; OUT(is, 5, j, i) = (float)r232;
                OUT(is, 6, i, j) = (float)(-r311); OUT(is, 6, j, i) = (float)(-r312);
                OUT(is, 7, i, j) = (float)(-r321); OUT(is, 7, j, i) = (float)(-r322);
                OUT(is, 8, i, j) = (float)(-r331); OUT(is, 8, j, i) = (float)(-r332);
            }
        }
    }
    free(alpha); free(beta); free(gamma); free(zeta); free(g); free(kb);
    return 0;
}
