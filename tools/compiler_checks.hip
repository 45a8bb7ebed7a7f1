// tools/compiler_checks.hip -- the device-side evidence behind three statements in DESIGN.md 4.1 / fsk_demod_wave.hip (round 2's scratch
// programs were lost with tools/scratch/; this file replaces swap_test.hip, sqrt_exhaustive.hip and sqrt_variants.hip):
//   1. __builtin_amdgcn_permlane32_swap under hipcc 7.2: do the two elements of its result differ as the ISA says, or does the
//      second read back as the first (why the Ndft = 512 instances use inline asm)?
//   2. v_sqrt_f32 against the correctly rounded square root over EVERY positive float: the histogram of its error in ulps
//      (why the estimator does not use the bare instruction: Sf has to be bit-identical to sqrtf on the CPU).
//   3. the kernel's replacement, q = min(rsq(x), 2^60); y = x q; fma(fma(-y, y, x), q/2, y), over x = 0 and [2^-96, FLT_MAX], and the
//      v_sqrt + neighbour-residual form: number of wrong results (the library's own self-test, pirip_hip_selftest_sqrt, repeats this).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/compiler_checks tools/compiler_checks.hip && tools/bin/compiler_checks > profiles/r04_compiler_checks.txt
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>

__global__ void swap_kernel(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    const unsigned a = 1000u + lane, b = 2000u + lane;
    // builtin: returns {vdst_new, vsrc_new}; lanes 0..31 of one operand are exchanged with lanes 32..63 of the other
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    unsigned x = a, y = b;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    out[lane] = r[0]; out[64 + lane] = r[1]; out[128 + lane] = x; out[192 + lane] = y;
}

__device__ __forceinline__ float sqrt_rsq(float x)
{
    float q = __builtin_amdgcn_rsqf(x);
    q = fminf(q, 0x1p60f);
    const float y = x * q, h = 0.5f * q;
    return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
}
__device__ __forceinline__ float sqrt_fix(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1), yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    float r = (rm <= 0.0f) ? ym : y;
    return (rp > 0.0f) ? yp : r;
}

// hist[0..4]: v_sqrt_f32 - RN(sqrt) in ulps (<= -2, -1, 0, +1, >= +2); hist[5], hist[6]: wrong results of the rsq form / the v_sqrt + fix-up form
__global__ void sqrt_kernel(unsigned long long *hist, unsigned lo, unsigned hi)
{
    unsigned long long h[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        const float want = (float)sqrt((double)x);                 // double rounding is innocuous for sqrt
        const int d = __builtin_bit_cast(int, __builtin_amdgcn_sqrtf(x)) - __builtin_bit_cast(int, want);
        h[d <= -2 ? 0 : d == -1 ? 1 : d == 0 ? 2 : d == 1 ? 3 : 4]++;
        h[5] += __builtin_bit_cast(unsigned, sqrt_rsq(x)) != __builtin_bit_cast(unsigned, want);
        h[6] += __builtin_bit_cast(unsigned, sqrt_fix(x)) != __builtin_bit_cast(unsigned, want);
    }
    for (int i = 0; i < 7; i++) if (h[i]) atomicAdd(&hist[i], h[i]);
}

static void run_sqrt(const char *name, unsigned lo, unsigned hi)
{
    unsigned long long *d, h[7];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(sqrt_kernel, dim3(4096), dim3(256), 0, 0, d, lo, hi);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipFree(d);
    const double n = (double)hi - (double)lo + 1.0;
    printf("%-28s v_sqrt_f32 - RN(sqrt) in ulps:  <=-2: %llu   -1: %llu (%.2f %%)   0: %llu   +1: %llu (%.4f %%)   >=+2: %llu | wrong: rsq + residual step %llu, "
           "v_sqrt + neighbour test %llu\n", name, h[0], h[1], 100.0 * h[1] / n, h[2], h[3], 100.0 * h[3] / n, h[4], h[5], h[6]);
}

int main()
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) { fprintf(stderr, "no HIP device\n"); return 2; }
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(swap_kernel, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipFree(d);
    // ISA: v_permlane32_swap vdst, vsrc exchanges vdst[32..63] with vsrc[0..31]: vdst' = {a[0..31], b[0..31]}, vsrc' = {a[32..63], b[32..63]}
    bool asm_ok = true, bi_ok = true, bi_same = true;
    for (int l = 0; l < 64; l++) {
        const unsigned wx = l < 32 ? 1000u + l : 2000u + (l - 32), wy = l < 32 ? 1000u + (l + 32) : 2000u + l;
        asm_ok &= h[128 + l] == wx && h[192 + l] == wy;
        bi_ok &= h[l] == wx && h[64 + l] == wy;
        bi_same &= h[l] == h[64 + l];
    }
    printf("# tools/compiler_checks.hip on this device / this hipcc\n");
    printf("permlane32_swap, inline asm : %s\n", asm_ok ? "as the ISA describes" : "UNEXPECTED");
    printf("permlane32_swap, builtin    : %s (element 0 lanes 0/32/63 = %u %u %u, element 1 = %u %u %u)\n",
           bi_ok ? "as the ISA describes" : bi_same ? "BOTH ELEMENTS READ BACK AS THE FIRST" : "differs from the ISA", h[0], h[32], h[63], h[64], h[96], h[127]);
    run_sqrt("all positive normals", 0x00800000u, 0x7f7fffffu);
    run_sqrt("2^-96 .. FLT_MAX", 0x0f800000u, 0x7f7fffffu);
    run_sqrt("x = 0", 0u, 0u);
    run_sqrt("denormals", 0x00000001u, 0x007fffffu);
    return 0;
}
