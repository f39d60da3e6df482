// csrc/land_models.h -- the land-surface model functions shared by the matrix generation (land.hip) and by the direct
// surface term of the azimuth recomposition (trphi.hip).  Compiled with fp contraction off by the including files.
#pragma once
#include <hip/hip_runtime.h>

// SOS_CALC_F_ROUJEAN (SOS_ROUJEAN.F:891-1000); phi in the Roujean convention
__device__ inline double calc_f_roujean(double k0, double k1, double k2, double c1, double s1, double c2, double s2, double phi)
{
    const double pi = 4. * atan(1.0);
    double xphi = phi;
    if (xphi < 0.) xphi = -xphi;
    if (xphi > pi) xphi = 2. * pi - xphi;
    double xc1 = c1, xs1 = s1, xc2 = c2, xs2 = s2;
    if (acos(c1) * 180. / pi > 60) { xc1 = cos(60 * pi / 180.); xs1 = sin(60 * pi / 180.); }     // CTE_TETAS_LIM_ROUJEAN
    if (acos(c2) * 180. / pi > 60) { xc2 = cos(60 * pi / 180.); xs2 = sin(60 * pi / 180.); }     // CTE_TETAV_LIM_ROUJEAN
    const double cosphi = cos(xphi), tants = xs1 / xc1, tantv = xs2 / xc2;
    double f1 = 0.5 * ((pi - xphi) * cosphi + sin(xphi)) * tants * tantv;
    f1 = f1 - tants - tantv;
    f1 = f1 - sqrt(tants * tants + tantv * tantv - 2. * tantv * tants * cosphi);
    f1 = f1 / pi;
    double coszeta = xc1 * xc2 + xs1 * xs2 * cosphi;
    if (fabs(fabs(coszeta) - 1.) <= 1.e-10) coszeta = (coszeta >= (1. - 1.e-10) && coszeta <= (1. + 1.e-10)) ? 1. : -1.;
    const double zeta = acos(coszeta);
    double f2 = 4. * ((pi / 2. - zeta) * coszeta + sin(zeta)) / (3. * pi * (xc1 + xc2));
    f2 = f2 - (1. / 3.);
    return (k0 + k1 * f1 + k2 * f2) * c2 * c1;
}

// SOS_CALCG_MAIGNAN (SOS_SURFACE_BPDF.F:1606-1641)
__device__ inline double calcg_maignan(double c1, double c2, double s12, double phi, double coef_c)
{
    const double cos2i = c1 * c2 - s12 * cos(phi);
    double tan2i = (1 - cos2i) / (1 + cos2i);
    if (tan2i < 0.) tan2i = 0.;
    return coef_c * exp(-sqrt(tan2i)) / (1. / c1 + 1. / c2);
}
