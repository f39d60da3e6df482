// Micro-benchmark: what does a wave pay for its vector / LDS instructions while the OTHER wave of its SIMD streams
// v_mfma_f64_16x16x4_f64?  One workgroup of 8 waves: waves 0-3 (one per SIMD) run the "load" (MFMA stream or idle),
// waves 4-7 (same SIMDs) run a probe loop of one instruction kind and time it with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_contention scripts/ubench_contention.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

enum Probe { P_F64_DEP, P_F64_IND, P_F32_IND, P_INT_IND, P_LDS_RD, P_LDS_RW, P_MFMA, P_NKIND };
static const char *kname[] = {"f64 fma dependent", "f64 fma independent x8", "f32 fma independent x8", "int add independent x8",
                              "lds read b64 x8", "lds read+write b64 x8", "mfma f64 16x16x4 x8"};

template <int KIND>
__device__ __forceinline__ unsigned long long probe(int iters, double *lds, double &sink)
{
    const int lane = threadIdx.x & 63;
    double a[8];
    float f[8];
    int n[8];
    v4d acc[8];
    for (int i = 0; i < 8; i++) { a[i] = 1.0 + i * 1e-3 + lane * 1e-6; f[i] = (float)a[i]; n[i] = i + lane; acc[i] = (v4d){0., 0., 0., 0.}; }
    const double m = 0.999999, c = 1e-9;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (KIND == P_F64_DEP) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[0] = a[0] * m + c;
        } else if (KIND == P_F64_IND) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = a[i] * m + c;
        } else if (KIND == P_F32_IND) {
#pragma unroll
            for (int i = 0; i < 8; i++) f[i] = f[i] * 0.99999f + 1e-6f;
        } else if (KIND == P_INT_IND) {
#pragma unroll
            for (int i = 0; i < 8; i++) n[i] = n[i] * 3 + it;
        } else if (KIND == P_LDS_RD) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += lds[lane + 64 * i + (it & 1)];
        } else if (KIND == P_LDS_RW) {
#pragma unroll
            for (int i = 0; i < 8; i++) { a[i] += lds[lane + 64 * i]; lds[lane + 64 * i] = a[i]; }
        } else if (KIND == P_MFMA) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[1], acc[i], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.;
    for (int i = 0; i < 8; i++) s += a[i] + f[i] + n[i] + acc[i][0] + acc[i][3];
    sink = s;
    return t1 - t0;
}

__global__ __launch_bounds__(512) void k_contention(int kind, int load, int iters, unsigned long long *out, double *sinkbuf, int *simd_ids, unsigned long long *lout)
{
    __shared__ double lds[4 * 1024];
    __shared__ int stop;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * 1024; i += 512) lds[i] = 1e-3 * i;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    if (lane == 0) simd_ids[wv] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_ID SIMD_ID[5:4]
    double sink = 0.;
    if (wv < 4) {
        // load waves: stream MFMAs (16 independent accumulators) until the probes are done
        if (load) {
            v4d acc[16];
            for (int i = 0; i < 16; i++) acc[i] = (v4d){0., 0., 0., 0.};
            double x = 1.0 + lane * 1e-6, y = 0.5;
            int guard = 0;
            const unsigned long long l0 = __builtin_amdgcn_s_memrealtime();
            while (!__builtin_amdgcn_readfirstlane(*(volatile int *)&stop) && guard < 4000000) {
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
                guard += 64;
            }
            const unsigned long long l1 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) { lout[2 * wv] = l1 - l0; lout[2 * wv + 1] = guard; }
            for (int i = 0; i < 16; i++) sink += acc[i][0];
        }
    } else {
        unsigned long long dt = 0;
        double *myl = lds + (wv - 4) * 1024;
        switch (kind) {
        case P_F64_DEP: dt = probe<P_F64_DEP>(iters, myl, sink); break;
        case P_F64_IND: dt = probe<P_F64_IND>(iters, myl, sink); break;
        case P_F32_IND: dt = probe<P_F32_IND>(iters, myl, sink); break;
        case P_INT_IND: dt = probe<P_INT_IND>(iters, myl, sink); break;
        case P_LDS_RD: dt = probe<P_LDS_RD>(iters, myl, sink); break;
        case P_LDS_RW: dt = probe<P_LDS_RW>(iters, myl, sink); break;
        case P_MFMA: dt = probe<P_MFMA>(iters, myl, sink); break;
        }
        if (lane == 0) out[wv - 4] = dt;
        __threadfence_block();
        if (lane == 0) atomicAdd(&stop, 1);
        // wait for all four probe waves, then release the load waves
        while (__builtin_amdgcn_readfirstlane(*(volatile int *)&stop) < 4) __builtin_amdgcn_s_sleep(2);
    }
    sinkbuf[threadIdx.x] = sink;
}

int main()
{
    unsigned long long *d_out;
    double *d_sink;
    int *d_simd;
    hipMalloc(&d_out, 4 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 512 * sizeof(double));
    hipMalloc(&d_simd, 8 * sizeof(int));
    unsigned long long *d_l;
    hipMalloc(&d_l, 8 * sizeof(unsigned long long));
    const int iters = 20000;
    int simd[8];
    printf("%-28s %14s %14s   (ns per probe instruction, s_memrealtime 100 MHz; MFMA alone = 27.7 ns = 64 cycles at 2.31 GHz)\n", "probe", "alone", "vs MFMA stream");
    for (int kind = 0; kind < P_NKIND; kind++) {
        double res[2];
        for (int load = 0; load < 2; load++) {
            hipLaunchKernelGGL(k_contention, dim3(1), dim3(512), 0, 0, kind, load, iters, d_out, d_sink, d_simd, d_l);
            hipDeviceSynchronize();
            unsigned long long o[4];
            hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
            hipMemcpy(simd, d_simd, sizeof(simd), hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 4; i++) avg += (double)o[i];
            res[load] = avg / 4 / ((double)iters * 8);
        }
        unsigned long long l[8];
        hipMemcpy(l, d_l, sizeof(l), hipMemcpyDeviceToHost);
        double lr = 0;
        for (int i = 0; i < 4; i++) lr += (double)l[2 * i] / (double)l[2 * i + 1];
        printf("%-28s %14.3f %14.3f   load waves: %.3f ns per MFMA\n", kname[kind], res[0] * 10, res[1] * 10, lr / 4 * 10);
    }
    printf("SIMD ids of waves 0..7:");
    for (int i = 0; i < 8; i++) printf(" %d", simd[i]);
    printf("\n");
    // s_memtime tick vs shader clock: time a known-length dependent chain
    return 0;
}
