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

// Orders a bin did not run (s >= norders[b]) are NOT read: the solver leaves them unwritten (the reference's shorter
// FICOS file, zero-padded by SOS_AGGREGATE.F:357-413), and a failed bin (norders < 0, the reference's IER = -1)
// contributes nothing -- it is reported through the scalar block instead.
__global__ void k_aggregate(int nel, int per_order, const int32_t *seg, const double *aik, const int32_t *norders,
                            const double *rec, double *out)
{
    const int g = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    const int s = e / per_order;
    const int b0 = seg[g], b1 = seg[g + 1];
    double acc = 0.;
    for (int b = b0; b < b1; b++)
        if (s < norders[b]) acc = acc + aik[b] * rec[(size_t)b * nel + e];
    out[(size_t)g * nel + e] = acc;
}

// Large single segments (throughput batches): bins are split into chunks accumulated serially and the chunk
// partials are then summed in chunk order -- deterministic, same operation per term as SOS_AGGREGATE.F:401-403,
// only the association of the outer sum differs from the strict serial order (<= 1e-16 relative).
__global__ void k_aggregate_partial(int nel, int per_order, int b0, int b1, int chunk, const double *aik,
                                    const int32_t *norders, const double *rec, double *partial)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    const int s = e / per_order;
    const int c = blockIdx.y;
    const int lo = b0 + c * chunk, hi = min(lo + chunk, b1);
    double acc = 0.;
#pragma unroll 8
    for (int b = lo; b < hi; b++)
        if (s < norders[b]) acc = acc + aik[b] * rec[(size_t)b * nel + e];
    partial[(size_t)c * nel + e] = acc;
}

__global__ void k_aggregate_final(int nel, int nchunk, const double *partial, double *out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    double acc = 0.;
    for (int c = 0; c < nchunk; c++) acc = acc + partial[(size_t)c * nel + e];
    out[e] = acc;
}

// out_scal[g][SOSGPU_SCAL_BASE + N] (include/sosgpu.h): sum aik*TDIFMUS, sum aik*EMOINS, sum aik*EPLUS,
// sum aik*exp(-TTOT_TRONC), sum aik*exp(-TTOT_VRAI), sum aik*exp(-TAUOUT), sum aik, max norders, -(min norders), 0,
// then sum aik*TDIFMUG(j), j = 1..N (SOS_AGGREGATE.F:452-459; zero when no per-bin TDIFMUG is given).
// Elements 7 and 8 combine across ranks with MAX, everything else with SUM.
#define SCB 10
__global__ void k_aggregate_scal(int n, const int32_t *seg, const double *aik, const int32_t *norders,
                                 const double *flux, const double *scal, const double *tdifmug, double *out)
{
    // 256 threads stride over the bins of the segment (serial within a thread), then a fixed-shape tree
    // over the 256 partials: deterministic for a given segment size.
    __shared__ double sm[256][SCB];
    const int g = blockIdx.x, t = threadIdx.x;
    const int b0 = seg[g], b1 = seg[g + 1];
    double a[SCB];
    for (int i = 0; i < SCB; i++) a[i] = 0.;
    a[8] = -2147483647.;
    for (int b = b0 + t; b < b1; b += 256) {
        const double w = aik[b];
        const int no = norders[b];
        a[7] = fmax(a[7], (double)no);
        a[8] = fmax(a[8], -(double)no);
        if (no < 0) continue;                       // failed bin: flagged through a[8], no contribution
        a[0] = a[0] + w * scal[4 * b + 0];
        a[1] = a[1] + w * flux[2 * b + 0];
        a[2] = a[2] + w * flux[2 * b + 1];
        a[3] = a[3] + w * exp(-scal[4 * b + 1]);
        a[4] = a[4] + w * exp(-scal[4 * b + 2]);
        a[5] = a[5] + w * exp(-scal[4 * b + 3]);
        a[6] = a[6] + w;
    }
    for (int i = 0; i < SCB; i++) sm[t][i] = a[i];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (t < st)
            for (int i = 0; i < SCB; i++)
                sm[t][i] = (i == 7 || i == 8) ? fmax(sm[t][i], sm[t + st][i]) : sm[t][i] + sm[t + st][i];
        __syncthreads();
    }
    const int sw = SCB + n;
    if (t < SCB) out[(size_t)sw * g + t] = sm[0][t];
    // TDIFMUG(j): one thread per direction, bins in serial order (SOS_AGGREGATE.F:455-458)
    for (int j = t; j < n; j += 256) {
        double acc = 0.;
        if (tdifmug)
            for (int b = b0; b < b1; b++)
                if (norders[b] >= 0) acc = acc + aik[b] * tdifmug[(size_t)b * n + j];
        out[(size_t)sw * g + SCB + j] = acc;
    }
}

void launch_aggregate(const SosDev &cx, int nseg, const int32_t *d_seg, const double *d_aik,
                      const double *d_rec, const int32_t *d_norders, const double *d_flux, const double *d_scal,
                      const double *d_tdifmug, double *d_out_rec, double *d_out_scal, hipStream_t st, int nb_single,
                      double *d_partial, int max_chunks)
{
    const int per_order = 3 * cx.w;
    const int nel = (cx.smax + 1) * per_order;
    const int chunk = 64;
    const int nchunk = (nb_single + chunk - 1) / chunk;
    if (nseg == 1 && nb_single > 2 * chunk && d_partial && nchunk <= max_chunks) {
        dim3 grid((nel + 255) / 256, nchunk);
        k_aggregate_partial<<<grid, 256, 0, st>>>(nel, per_order, 0, nb_single, chunk, d_aik, d_norders, d_rec, d_partial);
        k_aggregate_final<<<(nel + 255) / 256, 256, 0, st>>>(nel, nchunk, d_partial, d_out_rec);
    } else {
        dim3 grid((nel + 255) / 256, nseg);
        k_aggregate<<<grid, 256, 0, st>>>(nel, per_order, d_seg, d_aik, d_norders, d_rec, d_out_rec);
    }
    k_aggregate_scal<<<nseg, 256, 0, st>>>(cx.n, d_seg, d_aik, d_norders, d_flux, d_scal, d_tdifmug, d_out_scal);
}
