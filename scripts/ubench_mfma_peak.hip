// Micro-benchmark: FP64 MFMA throughput of gfx950 as a function of resident waves per SIMD, whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_mfma_peak scripts/ubench_mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k_mfma(int iters, double *sink, unsigned long long *ticks)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0., 0., 0., 0.};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

int main()
{
    double *d_sink;
    unsigned long long *d_t;
    hipMalloc(&d_sink, (size_t)256 * 8 * 1024 * sizeof(double));
    hipMalloc(&d_t, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, NACC = 8;
    printf("waves/SIMD  grid  ms      TFLOP/s(chip)  ticks/MFMA(wave)  tick GHz\n");
    for (int wps = 1; wps <= 4; wps++) {
        const int threads = 256 * wps;            // 4 SIMDs x wps waves x 64 lanes, one workgroup per CU
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma<NACC>, dim3(256), dim3(threads), 0, 0, iters, d_sink, d_t);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t;
            hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
            const double flops = 256.0 * (threads / 64) * (double)iters * NACC * 2048.0;
            if (rep) printf("%d           256   %.3f  %.1f           %.2f             %.3f\n", wps, ms, flops / (ms * 1e-3) / 1e12,
                            (double)t / ((double)iters * NACC), (double)t / (ms * 1e6));
        }
    }
    return 0;
}
