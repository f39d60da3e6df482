// Layout probe of v_mfma_f64_4x4x4f64 on gfx950: for every (A lane, B lane) pair put 1.0 in those two lanes only and
// record which output lanes become non-zero.  Prints the block / row / k / column mapping it implies.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (c != 0.0) out[la * 64 + lb] = lane;          // at most one output element per pair
        }
}
int main()
{
    int *d, h[4096];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0xff, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // for each A lane: list of (B lane -> out lane)
    for (int la = 0; la < 64; la++) {
        printf("A%2d:", la);
        for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf(" B%d->D%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
