// Diagnostic: which wave slots (HW_ID.WAVE_ID) do the waves of two co-resident 4-wave workgroups of a CU get?
// hipcc --offload-arch=gfx950 -O2 scripts/probe_waveslot.hip -o scripts/probe_waveslot && scripts/probe_waveslot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int spin)
{
    extern __shared__ double smem[];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + wv) * 2] = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | 4);        // HW_ID[15:0]
        out[(blockIdx.x * 4 + wv) * 2 + 1] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20);    // XCC_ID
    }
    smem[threadIdx.x] = 1.0;
    for (int i = 0; i < spin; i++) { __syncthreads(); smem[threadIdx.x] += smem[(threadIdx.x + 1) & 255]; }
    if (smem[threadIdx.x] == 12345.0) out[0] = 0;
}
int main()
{
    const int nb = 512 * 3;
    unsigned *d;
    hipMalloc(&d, nb * 8 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024);
    k<<<nb, 256, 81 * 1024>>>(d, 20000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> slots, simd_of_wave[4];
    int same = 0, mixed = 0;
    for (int b = 0; b < nb; b++) {
        unsigned s0 = h[(b * 4) * 2] & 15;
        bool all = true;
        for (int w = 0; w < 4; w++) {
            unsigned id = h[(b * 4 + w) * 2];
            slots[id & 15]++;
            simd_of_wave[w][(id >> 4) & 3]++;
            if ((id & 15) != s0) all = false;
        }
        (all ? same : mixed)++;
    }
    printf("WAVE_ID histogram:"); for (auto &p : slots) printf(" %u:%d", p.first, p.second); printf("\n");
    for (int w = 0; w < 4; w++) { printf("wave %d SIMD histogram:", w); for (auto &p : simd_of_wave[w]) printf(" %u:%d", p.first, p.second); printf("\n"); }
    printf("workgroups whose 4 waves share one WAVE_ID: %d, mixed: %d\n", same, mixed);
    return 0;
}
