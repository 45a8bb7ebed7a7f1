// tools/scratch/sqrt_variants.hip -- exhaustive check of cheaper correctly-rounded sqrt candidates against (float)sqrt((double)x)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int V> __device__ float cand(float x)
{
    if (V == 0) {   // shipped: v_sqrt + neighbour residual test
        const float y = __builtin_amdgcn_sqrtf(x);
        const float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1), yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
        const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
        float r = (rm <= 0.0f) ? ym : y; r = (rp > 0.0f) ? yp : r; return r;
    } else if (V == 1) {
        const float y = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    } else if (V == 2) {
        const float y = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rcpf(y);
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    } else if (V == 3) {
        const float s = __builtin_amdgcn_rsqf(x), y = x * s, h = 0.5f * s;
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    } else if (V == 4) {
        const float s = __builtin_amdgcn_rsqf(x), h = 0.5f * s;
        float y = x * s;
        y = __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    } else if (V == 6) {   // candidate to ship: variant 3 with the reciprocal root clamped so that x = 0 gives 0
        float q = __builtin_amdgcn_rsqf(x);
        q = __builtin_fminf(q, 0x1p60f);
        const float y = x * q, h = 0.5f * q;
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    } else {   // LLVM's flush-denormal expansion
        const float s = __builtin_amdgcn_rsqf(x);
        float g = x * s, h = 0.5f * s;
        const float e = __builtin_fmaf(-h, g, 0.5f);
        h = __builtin_fmaf(h, e, h); g = __builtin_fmaf(g, e, g);
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
}
template <int V> __global__ void k(unsigned long long *bad, unsigned *first, unsigned lo, unsigned hi)
{
    unsigned long long c = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        if (__builtin_bit_cast(unsigned, cand<V>(x)) != __builtin_bit_cast(unsigned, (float)sqrt((double)x))) { c++; atomicMin(first, (unsigned)b); }
    }
    if (c) atomicAdd(bad, c);
}
template <int V> void run(const char *name, unsigned long long *d, unsigned *f, unsigned lo = 0x0f800000u, unsigned hi = 0x5f000000u)
{
    hipMemset(d, 0, 8); hipMemset(f, 0xff, 4);
    hipLaunchKernelGGL(k<V>, dim3(4096), dim3(256), 0, 0, d, f, lo, hi);
    unsigned long long h; unsigned ff;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&ff, f, 4, hipMemcpyDeviceToHost);
    printf("%-44s wrong results on [2^-96, 2^63]: %llu (first at bits 0x%08x)\n", name, h, ff);
}
int main()
{
    unsigned long long *d; unsigned *f;
    hipMalloc(&d, 8); hipMalloc(&f, 4);
    run<0>("v_sqrt + neighbour residual test (shipped)", d, f);
    run<1>("v_sqrt, r=fma(-y,y,x), fma(r, .5*rsq(x), y)", d, f);
    run<2>("v_sqrt, r=fma(-y,y,x), fma(r, .5*rcp(y), y)", d, f);
    run<3>("y=x*rsq, one residual step", d, f);
    run<4>("y=x*rsq, two residual steps", d, f);
    run<5>("rsq + Goldschmidt (LLVM ftz expansion)", d, f);
    run<6>("clamped variant 3, [2^-96, 2^63]", d, f);
    run<6>("clamped variant 3, zero and 2^-96 .. FLT_MAX", d, f, 0x0f800000u, 0x7f7fffffu);
    run<6>("clamped variant 3, x = 0 only", d, f, 0u, 0u);
    run<6>("clamped variant 3, all normals", d, f, 0x00800000u, 0x7f7fffffu);
    run<6>("clamped variant 3, denormals", d, f, 1u, 0x007fffffu);
    run<3>("variant 3, all normals", d, f, 0x00800000u, 0x7f7fffffu);
    return 0;
}
