// csrc/aggregate.hip -- CKD-bin aggregation (gfx950).
//
// Replaces SOS_AGGREGATE (reference src/SOS_AGGREGATE.F:372-488), which re-reads and re-writes the
// running result file once per bin.  Here the per-bin Fourier records stay in HBM and one kernel forms
//     out[g][s][c][j] = sum_{b in segment g} AIK(b) * rec[b][s][c][j]
// for every wavelength/band segment g at once.  Bins are accumulated serially IN BIN ORDER by each
// thread (same order and same un-fused multiply-add as SOS_AGGREGATE.F:401-403), so a single-GPU result
// is bit-identical to the serial reference; reads are coalesced across threads (consecutive elements).
// Pure HBM streaming: nb*(smax+1)*3*W*8 bytes read once.
// The three optical depths are combined as sum AIK*exp(-tau) (SOS_AGGREGATE.F:467-488); -ln is applied
// on the host after the cross-GPU reduce.
#include "sos_common.h"
#include "kernels.h"

#pragma clang fp contract(off)

__global__ void k_aggregate(int nel, const int32_t *seg, const double *aik, const double *rec, double *out)
{
    const int g = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    const int b0 = seg[g], b1 = seg[g + 1];
    double acc = 0.;
    for (int b = b0; b < b1; b++) acc = acc + aik[b] * rec[(size_t)b * nel + e];
    out[(size_t)g * nel + e] = acc;
}

// out_scal[g][8]: sum aik*TDIFMUS, sum aik*EMOINS, sum aik*EPLUS, sum aik*exp(-TTOT_TRONC),
//                 sum aik*exp(-TTOT_VRAI), sum aik*exp(-TAUOUT), sum aik, max norders
__global__ void k_aggregate_scal(const int32_t *seg, const double *aik, const int32_t *norders,
                                 const double *flux, const double *scal, double *out)
{
    const int g = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    for (int b = seg[g]; b < seg[g + 1]; b++) {
        const double w = aik[b];
        a[0] = a[0] + w * scal[4 * b + 0];
        a[1] = a[1] + w * flux[2 * b + 0];
        a[2] = a[2] + w * flux[2 * b + 1];
        a[3] = a[3] + w * exp(-scal[4 * b + 1]);
        a[4] = a[4] + w * exp(-scal[4 * b + 2]);
        a[5] = a[5] + w * exp(-scal[4 * b + 3]);
        a[6] = a[6] + w;
        a[7] = fmax(a[7], (double)norders[b]);
    }
    for (int i = 0; i < 8; i++) out[8 * g + i] = a[i];
}

void launch_aggregate(const SosDev &cx, int nseg, const int32_t *d_seg, const double *d_aik,
                      const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                      double *d_out_rec, double *d_out_scal, hipStream_t st)
{
    const int nel = (cx.smax + 1) * 3 * cx.w;
    dim3 grid((nel + 255) / 256, nseg);
    k_aggregate<<<grid, 256, 0, st>>>(nel, d_seg, d_aik, d_rec, d_out_rec);
    k_aggregate_scal<<<nseg, 64, 0, st>>>(d_seg, d_aik, d_norders, d_flux, d_scal, d_out_scal);
}
