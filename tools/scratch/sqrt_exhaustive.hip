// tools/scratch/sqrt_exhaustive.hip -- how does gfx950's v_sqrt_f32 err?  For every positive normal float: difference in ulps
// between v_sqrt_f32 and the correctly rounded root ((float)sqrt((double)x): double rounding is innocuous for sqrt).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/sqrt_exhaustive tools/scratch/sqrt_exhaustive.hip && tools/bin/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(unsigned long long *hist, unsigned lo, unsigned hi)
{
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        const float y = __builtin_amdgcn_sqrtf(x);
        const float r = (float)sqrt((double)x);
        int d = __builtin_bit_cast(int, y) - __builtin_bit_cast(int, r);
        d = d < -2 ? -2 : d > 2 ? 2 : d;
        cnt[d + 2]++;
    }
    for (int i = 0; i < 5; i++) if (cnt[i]) atomicAdd(&hist[i], cnt[i]);
}
int main()
{
    unsigned long long *d, h[5];
    hipMalloc(&d, sizeof(h));
    struct { unsigned lo, hi; const char *name; } rg[] = {
        {0x00800000u, 0x7f7fffffu, "all positive normals"},
        {0x0f800000u, 0x5f000000u, "2^-96 .. 2^63"},
        {0x00000001u, 0x007fffffu, "denormals"},
    };
    for (auto &g : rg) {
        hipMemset(d, 0, sizeof(h));
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, g.lo, g.hi);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-24s: v_sqrt_f32 - RN(sqrt) in ulps:  <=-2: %llu   -1: %llu   0: %llu   +1: %llu   >=+2: %llu\n", g.name, h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
