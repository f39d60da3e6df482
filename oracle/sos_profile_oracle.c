/* oracle/sos_profile_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or imported by the product path).
 *
 * Plain-C restatement of the reference's atmospheric profile discretisation for IPROFIL = 1 (exponential aerosol
 * profile), with or without gas absorption:
 *   SOS_PROFILE   src/SOS_PROFIL.F:224-1170   (no-gas step :349-489, selection :492-508, gas step :509-795,
 *                                               PROFIL file written with format 20 `2X,I5,F10.5,3(E15.8)` :1084,1150)
 *   SOS_DISC      src/SOS_PROFIL.F:1210-1332  (bisection on altitude)
 * Statement order follows the Fortran; REAL*4 literals of inc/SOS.h (CTE_TCOUCHE 0.005, first-layer thickness 0.0002,
 * CTE_DELTA_Z 0.05, CTE_THRESHOLD_DZ 0.001, the .000001 of SOS_DISC) are widened from float exactly as Fortran does.
 * The returned arrays are the values SOS reads back from the PROFIL file (SOS.F:511-516): ZPROF through F10.5, H,
 * PCAER (= XDEL), PCMOL (= YDEL) through E15.8.
 * Pinned against the real reference (oracle/_ref, sos_profile_) by tests/golden/profile_*.npz.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OS_NT 600          /* CTE_OS_NT            SOS.h:202 */
#define OS_NT_MIN 100      /* CTE_OS_NT_MIN        SOS.h:229 */
#define ABS_NBLEV 50       /* CTE_ABS_NBLEV        SOS.h:250 */
#define TOA_ALT 120.0      /* CTE_TOA_ALT          SOS.h:197 */
#define THRESHOLD_TAUABS 1.5   /* SOS.h:301 (exact in REAL*4) */
static const double TCOUCHE = (double)0.005f;        /* SOS.h:208 */
static const double T_FIRST = (double)0.0002f;       /* CTE_TOA_FIRST_LAYER_OPT_THICKNESS  SOS.h:213 */
static const double DELTA_Z = (double)0.05f;         /* SOS.h:218 */
static const double THRESHOLD_DZ = (double)0.001f;   /* SOS.h:224 */

/* gas optical depth at altitude z by the linear interpolation of SOS_DISC (SOS_PROFIL.F:1283-1296) */
static double disc(double dt, double ta, double ha, double tr, double hr, const double *tabs, const double *altabs,
                   double tim1, double zmax_init, double tg_zlim, double zlim)
{
    const double ti = tim1 + dt;
    double zmax = zmax_init, zmin = zlim, zmoy;
    for (;;) {
        double tg;
        zmoy = (zmax + zmin) / 2.;
        if (tg_zlim > 0.0) {
            int j = 2;                                   /* 1-based */
            while (zmoy < altabs[j - 1]) j++;
            double zz;
            if (zmoy > altabs[0]) zz = 0;
            else zz = (zmoy - altabs[j - 2]) / (altabs[j - 1] - altabs[j - 2]);
            tg = (1 - zz) * tabs[j - 2] + zz * tabs[j - 1];
        } else tg = 0.0;
        const double tzmoy = ta * exp(-zmoy / ha) + tr * exp(-zmoy / hr) + tg;
        const double xd = fabs(ti - tzmoy);
        if (xd < (double).000001f) break;
        if (zmoy == 0.0) break;
        if ((ti - tzmoy) < 0.0) zmin = zmoy; else zmax = zmoy;
    }
    return zmoy;
}

static double rt_e15_8(double v) { char b[64]; snprintf(b, sizeof b, "%.7E", v); return strtod(b, NULL); }
static double rt_f10_5(double v) { char b[64]; snprintf(b, sizeof b, "%.5f", v); return strtod(b, NULL); }

/* returns 0, or -1 (IER) when the profile needs more than CTE_OS_NT levels / bad constants.
 * tabs == NULL or absprofil == 7 or tabs[49] == 0: no gas.  Arrays hold OS_NT+1 doubles. */
int sos_profile_oracle(double tr, double hr, double ta, double ha, int absprofil, const double *altabs,
                       const double *tabs, int *nt_out, double *zprof, double *h, double *pcaer, double *pcmol)
{
    static double hmol_ng[OS_NT + 2], haer_ng[OS_NT + 2], h_ng[OS_NT + 2], z_ng[OS_NT + 2], pcm_ng[OS_NT + 2], pca_ng[OS_NT + 2];
    static double hmol[OS_NT + 2], haer[OS_NT + 2], habs[OS_NT + 2];
    int nt_ng, nt, i;
    double t_first_ng, t_layer_ng, ttot, z, vr, va, vg, dtau;

    /* ---- step 1: profile without gas (SOS_PROFIL.F:349-489) */
    ttot = tr + ta;
    if ((ttot / OS_NT_MIN) <= T_FIRST) {
        nt_ng = OS_NT_MIN; t_layer_ng = ttot / nt_ng; t_first_ng = t_layer_ng;
    } else if ((ttot / OS_NT_MIN) < TCOUCHE) {
        nt_ng = OS_NT_MIN + 1; t_first_ng = T_FIRST; t_layer_ng = (ttot - t_first_ng) / OS_NT_MIN;
    } else {
        t_first_ng = T_FIRST;
        nt_ng = (int)((ttot - t_first_ng) / TCOUCHE);
        t_layer_ng = (ttot - t_first_ng) / nt_ng;
        nt_ng = nt_ng + 1;
    }
    if (nt_ng > OS_NT) return -1;
    if (ta == 0.0) {
        hmol_ng[0] = 0.; hmol_ng[1] = t_first_ng;
        for (i = 2; i <= nt_ng; i++) hmol_ng[i] = (i - 1) * t_layer_ng + t_first_ng;
        for (i = 0; i <= nt_ng; i++) { pcm_ng[i] = 1.; pca_ng[i] = 0.; haer_ng[i] = 0.; h_ng[i] = hmol_ng[i]; }
        z_ng[0] = TOA_ALT;
        for (i = 1; i <= nt_ng; i++) z_ng[i] = hr * log(tr / hmol_ng[i]);
    } else {
        z_ng[0] = TOA_ALT; hmol_ng[0] = 0.; haer_ng[0] = 0.; h_ng[0] = 0.;
        dtau = 0.; z = TOA_ALT;
        while (dtau < t_first_ng) { z = z - DELTA_Z; dtau = tr * exp(-z / hr) + ta * exp(-z / ha); }
        z_ng[1] = z;
        vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
        hmol_ng[1] = vr; haer_ng[1] = va; h_ng[1] = dtau;
        pcm_ng[1] = vr / dtau; pca_ng[1] = va / dtau;
        pcm_ng[0] = pcm_ng[1]; pca_ng[0] = pca_ng[1];
        for (i = 2; i <= nt_ng - 1; i++) {
            z = disc(t_layer_ng, ta, ha, tr, hr, tabs, altabs, h_ng[i - 1], z_ng[1], 0., 0.);
            z_ng[i] = z;
            vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
            hmol_ng[i] = vr; haer_ng[i] = va; h_ng[i] = vr + va;
            vr = vr - hmol_ng[i - 1]; va = va - haer_ng[i - 1];
            pcm_ng[i] = vr / (vr + va); pca_ng[i] = va / (vr + va);
        }
        z_ng[nt_ng] = 0.; hmol_ng[nt_ng] = tr; haer_ng[nt_ng] = ta; h_ng[nt_ng] = tr + ta;
        vr = tr - hmol_ng[nt_ng - 1]; va = ta - haer_ng[nt_ng - 1];
        pcm_ng[nt_ng] = vr / (vr + va); pca_ng[nt_ng] = va / (vr + va);
    }

    if (absprofil == 7 || tabs == NULL || tabs[ABS_NBLEV - 1] == 0.0) {     /* SOS_PROFIL.F:492-508 */
        nt = nt_ng;
        for (i = 0; i <= nt; i++) {
            zprof[i] = z_ng[i]; h[i] = hmol_ng[i] + haer_ng[i]; pcaer[i] = pca_ng[i]; pcmol[i] = pcm_ng[i];
        }
    } else {
        /* ---- step 2: profile with gas absorption (SOS_PROFIL.F:509-795) */
        double t_first, t_layer, zlim, tg_zlim, ttot_zlim, zing;
        int ing, j;
        const int strong = tabs[ABS_NBLEV - 1] > THRESHOLD_TAUABS;
        if (TCOUCHE > THRESHOLD_TAUABS) return -1;
        ttot = tr + ta + tabs[ABS_NBLEV - 1];
        if (strong) {
            i = 1;
            while (tabs[i - 1] < THRESHOLD_TAUABS) i++;
            const double alin = (tabs[i - 1] - tabs[i - 2]) / (altabs[i - 1] - altabs[i - 2]);
            const double blin = tabs[i - 1] - alin * altabs[i - 1];
            tg_zlim = THRESHOLD_TAUABS;
            zlim = (tg_zlim - blin) / alin;
            t_first = T_FIRST;
            ttot_zlim = ta * exp(-zlim / ha) + tr * exp(-zlim / hr) + tg_zlim;
            t_layer = (ttot_zlim - t_first) / (OS_NT - nt_ng - 2);
            t_layer = t_layer > TCOUCHE ? t_layer : TCOUCHE;
        } else {
            zlim = 0.; tg_zlim = tabs[ABS_NBLEV - 1];
            if ((ttot / OS_NT_MIN) <= T_FIRST) { nt = OS_NT_MIN; t_layer = ttot / nt; t_first = t_layer; }
            else if ((ttot / OS_NT_MIN) < TCOUCHE) { nt = OS_NT_MIN + 1; t_first = T_FIRST; t_layer = (ttot - t_first) / OS_NT_MIN; }
            else { t_first = T_FIRST; nt = (int)((ttot - t_first) / TCOUCHE); t_layer = (ttot - t_first) / nt; nt = nt + 1; }
        }
        nt = 1; z = TOA_ALT; zing = z_ng[1];
        hmol[0] = 0.; haer[0] = 0.; habs[0] = 0.; h[0] = 0.;
        ing = 1;
        ttot_zlim = ta * exp(-zlim / ha) + tr * exp(-zlim / hr) + tg_zlim;
        while ((ttot_zlim - h[nt - 1]) > t_layer) {
            i = nt;
            if (i > OS_NT - 1) return -1;                 /* the Fortran would overrun its arrays here */
            if (i == 1) {
                dtau = 0.;
                while (dtau < t_first) {
                    z = z - DELTA_Z;
                    j = 2;
                    while (z < altabs[j - 1]) j++;
                    if (z <= altabs[0]) {
                        const double zz = (z - altabs[j - 2]) / (altabs[j - 1] - altabs[j - 2]);
                        vg = (1 - zz) * tabs[j - 2] + zz * tabs[j - 1];
                    } else vg = 0.;
                    vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
                    dtau = vr + va + vg;
                }
                zprof[1] = z; h[1] = dtau; ing = 1;
            } else {
                z = disc(t_layer, ta, ha, tr, hr, tabs, altabs, h[i - 1], zprof[1], tg_zlim, zlim);
            }
            if (z <= zing) { z = zing; ing = ing + 1; zing = z_ng[ing]; }
            else if ((z - zing) <= THRESHOLD_DZ) { ing = ing + 1; zing = z_ng[ing]; }
            zprof[i] = z;
            j = 2;
            while (z < altabs[j - 1]) j++;
            if (z > altabs[0]) vg = tabs[j - 2];
            else {
                const double zz = (z - altabs[j - 2]) / (altabs[j - 1] - altabs[j - 2]);
                vg = (1 - zz) * tabs[j - 2] + zz * tabs[j - 1];
            }
            vr = tr * exp(-z / hr); va = ta * exp(-z / ha);
            hmol[i] = vr; haer[i] = va; habs[i] = vg;
            h[i] = va + vr + vg;
            va = va - haer[i - 1]; vr = vr - hmol[i - 1]; vg = vg - habs[i - 1];
            pcaer[i] = va / (va + vr + vg); pcmol[i] = vr / (va + vr + vg);
            nt = nt + 1;
        }
        if ((zprof[nt - 1] - zlim) <= THRESHOLD_DZ) nt = nt - 1;
        zprof[nt] = zlim;
        vr = tr * exp(-zlim / hr); va = ta * exp(-zlim / ha); vg = tg_zlim;
        hmol[nt] = vr; haer[nt] = va; habs[nt] = vg; h[nt] = vr + va + tg_zlim;
        va = va - haer[nt - 1]; vr = vr - hmol[nt - 1]; vg = vg - habs[nt - 1];
        pcaer[nt] = va / (va + vr + vg); pcmol[nt] = vr / (va + vr + vg);
        zprof[0] = TOA_ALT; pcaer[0] = pcaer[1]; pcmol[0] = pcmol[1];
        hmol[0] = 0.; haer[0] = 0.; habs[0] = 0.; h[0] = 0.;
        if (strong) {
            nt = nt + 1;
            if (nt > OS_NT) return -1;
            hmol[nt] = tr; haer[nt] = ta; habs[nt] = tabs[ABS_NBLEV - 1];
            h[nt] = hmol[nt] + haer[nt] + habs[nt];
            vr = hmol[nt] - hmol[nt - 1]; va = haer[nt] - haer[nt - 1]; vg = habs[nt] - habs[nt - 1];
            pcaer[nt] = va / (va + vr + vg); pcmol[nt] = vr / (va + vr + vg);
            zprof[nt] = 0.;
        }
    }
    /* PROFIL file round trip (write SOS_PROFIL.F:1084 format 20, read SOS.F:515 format 70) */
    for (i = 0; i <= nt; i++) {
        zprof[i] = rt_f10_5(zprof[i]); h[i] = rt_e15_8(h[i]); pcaer[i] = rt_e15_8(pcaer[i]); pcmol[i] = rt_e15_8(pcmol[i]);
    }
    *nt_out = nt;
    return 0;
}
