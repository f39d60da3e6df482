// Micro-benchmark for the "cluster" form of the streamed solver proposed in VERDICT r02 (item 2): C co-resident workgroups
// share one bin, each keeps one 32-level slab of the field in LDS and, once per scattering order, hands the other members its
// link vector (2 KB: KHM = 128 rows x 2 sweep directions) through L2 and waits for theirs.  What does that hand-off cost on
// gfx950?  Every workgroup: [optional compute phase of `work` back-to-back FP64 MFMAs per wave] -> publish 2 KB + release a
// flag -> wait until the C-1 other members' flags show this round -> acquire, read their 2 KB each.  512 workgroups of 256
// threads (two per CU, like k_sos_stream<4,2>), clusters either inside one XCD (members share blockIdx % 8: the hardware deals
// workgroups to the XCDs round-robin) or spread over the XCDs.  Time per round from s_memrealtime (100 MHz) of workgroup 0 and
// from HIP events over the launch.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_wg_exchange scripts/ubench_wg_exchange.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k_exchange(int C, int rounds, int same_xcd, int work, int fence, double *slots, int *flags,
                                                     unsigned long long *ticks, double *sink, int *xcd_of, int *abort_flag)
{
    const int b = blockIdx.x, t = threadIdx.x;
    int cluster, member;
    if (same_xcd) { const int q = b / 8; cluster = (b % 8) + 8 * (q / C); member = q % C; }
    else { cluster = b / C; member = b % C; }
    const int ncl_full = same_xcd ? 8 * ((int)gridDim.x / 8 / C) : (int)gridDim.x / C;
    if (cluster >= ncl_full) return;                                  // left-over workgroups (no complete cluster)
    if (t == 0) xcd_of[b] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7;   // HW_REG_XCC_ID
    double *myslot = slots + ((size_t)cluster * C + member) * 2 * 256;
    int *cf = flags + (size_t)cluster * 64;
    v4d acc = {0., 0., 0., 0.};
    double got = 0.;
    const double x = 1.0 + t * 1e-9, y = 0.5;
    __shared__ unsigned long long s_wait;
    if (t == 0) s_wait = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; r++) {
        for (int i = 0; i < work; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
        myslot[(r & 1) * 256 + t] = (double)(r + 1000 * member) + (work ? acc[0] * 0. : 0.);   // the link vector of this round
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&cf[member], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        if (t < C && t != member) {
            // (bounded: a member that is not resident -- fewer workgroup slots than the grid -- must not hang the device;
            //  flags[last] is the abort flag every waiter watches)
            long spins = 0;
            while (__hip_atomic_load(&cf[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (t == 0) s_wait += __builtin_amdgcn_s_memrealtime() - w0;
        for (int m = 0; m < C; m++)
            if (m != member) got += __hip_atomic_load(&slots[((size_t)cluster * C + m) * 2 * 256 + (r & 1) * 256 + t], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // every member must have seen every other member's vector of every round: sum_r sum_{m != member} (r + 1000 m)
    double expect = 0.;
    for (int m = 0; m < C; m++) if (m != member) expect += (double)rounds * (rounds + 1) / 2. + 1000. * m * rounds;
    if (got != expect && !__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicAdd(abort_flag + 1, 1);
    sink[(size_t)b * 256 + t] = got + acc[1];
    if (t == 0) { ticks[2 * b] = t1 - t0; ticks[2 * b + 1] = s_wait; }
}

int main()
{
    const int G = 512, rounds = 2000;
    double *d_slots, *d_sink;
    int *d_flags, *d_xcd;
    unsigned long long *d_t;
    hipMalloc(&d_slots, (size_t)G * 2 * 256 * sizeof(double));
    hipMalloc(&d_sink, (size_t)G * 256 * sizeof(double));
    hipMalloc(&d_flags, ((size_t)G * 64 + 16) * sizeof(int));
    hipMalloc(&d_xcd, G * sizeof(int));
    hipMalloc(&d_t, (size_t)2 * G * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("C  placement   sync      compute/round   us/round(events)  us/round(wg mean)  wait us/round (mean, max over wgs)  clusters on one XCD\n");
    const int works[3] = {0, 480, 1900};         // back-to-back MFMAs per wave and round: ~0, ~5 and ~20 us with two waves per SIMD
    for (int fence = 1; fence >= 0; fence--)
    for (int C : {2, 7, 13})
        for (int same = 1; same >= 0; same--)
            for (int wi = 0; wi < 3; wi++) {
                hipMemset(d_flags, 0, ((size_t)G * 64 + 16) * sizeof(int));
                hipMemset(d_xcd, 0xff, G * sizeof(int));
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_exchange, dim3(G), dim3(256), 0, 0, C, rounds, same, works[wi], fence, d_slots, d_flags, d_t, d_sink, d_xcd, d_flags + (size_t)G * 64);
                hipEventRecord(e1);
                if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                int aborted = 0;
                hipMemcpy(&aborted, d_flags + (size_t)G * 64, 4, hipMemcpyDeviceToHost);
                if (aborted) { printf("C = %d: a wait timed out (not all members resident) -- skipped\n", C); continue; }
                int wrong = 0;
                hipMemcpy(&wrong, d_flags + (size_t)G * 64 + 1, 4, hipMemcpyDeviceToHost);
                std::vector<unsigned long long> tk(2 * G);
                std::vector<int> xc(G);
                hipMemcpy(tk.data(), d_t, tk.size() * 8, hipMemcpyDeviceToHost);
                hipMemcpy(xc.data(), d_xcd, G * 4, hipMemcpyDeviceToHost);
                double sum = 0., wsum = 0., wmax = 0.;
                int n = 0, mono = 0, ncl = 0;
                for (int b = 0; b < G; b++) {
                    if (xc[b] < 0) continue;
                    sum += tk[2 * b]; wsum += tk[2 * b + 1]; if ((double)tk[2 * b + 1] > wmax) wmax = (double)tk[2 * b + 1]; n++;
                }
                // how many clusters really sit on one XCD
                const int nclf = same ? 8 * (G / 8 / C) : G / C;
                for (int c = 0; c < nclf; c++) {
                    int first = -1, ok = 1;
                    for (int m = 0; m < C; m++) {
                        const int b = same ? (c % 8) + 8 * ((c / 8) * C + m) : c * C + m;
                        if (first < 0) first = xc[b]; else if (xc[b] != first) ok = 0;
                    }
                    mono += ok; ncl++;
                }
                printf("%-2d %-11s %-9s %5d MFMA/wave   %8.2f          %8.2f           %8.2f  %8.2f       %d of %d   %s\n", C,
                       same ? "same XCD" : "across XCDs", fence ? "fences" : "no fences", works[wi], 1e3 * ms / rounds, sum / n / rounds * 0.01, wsum / n / rounds * 0.01,
                       wmax / rounds * 0.01, mono, ncl, wrong ? "WRONG DATA SEEN" : "data ok");
            }
    return 0;
}
