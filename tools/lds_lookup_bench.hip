// tools/lds_lookup_bench.hip -- what a data-dependent table look-up costs on the LDS pipe of one CU (gfx950), per wave-instruction:
//   lin      ds_read_b32, lane-linear addresses (conflict-free)
//   rnd576   ds_read_b32, addresses spread like the LDPC decoder's phi look-ups (576-entry table, pseudo-random index per lane)
//   bperm    ds_bpermute_b32 with a pseudo-random source lane (a 64-entry table held in one VGPR)
//   rnd64    ds_read_b32 on a 64-entry table (same addresses as bperm would use)
// 16 waves per CU (4 workgroups of 256), every wave issues the same instruction stream; cycles = elapsed * sclk / instructions per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_lookup_bench tools/lds_lookup_bench.hip && ./lds_lookup_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kIters = 2048, kUnroll = 8;

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float *out, const float *tab)
{
    __shared__ float s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float acc = 0.f;
    float reg = tab[lane];
    for (int it = 0; it < kIters; it++) {
        uint32_t idx[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            h = h * 1664525u + 1013904223u;
            idx[u] = h >> 8;
        }
        float v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            if (MODE == 0) v[u] = s[(threadIdx.x + u * 64 + (it & 3)) & 2047];
            else if (MODE == 1) v[u] = s[idx[u] % 576u];
            else if (MODE == 2) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((idx[u] & 63u) << 2), __builtin_bit_cast(int, reg)));
            else v[u] = s[idx[u] & 63u];
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) acc += v[u];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static double run(const char *name, float *out, const float *tab, int ncu)
{
    const int grid = ncu * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, tab);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, tab);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 16.0 * kIters * kUnroll;                  // wave-instructions of the measured kind per CU
    printf("%-8s %8.3f ms  %6.2f ns per wave-instruction per CU  (= %5.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *out, *tab;
    hipMalloc(&out, sizeof(float) * 256 * 4 * ncu);
    hipMalloc(&tab, sizeof(float) * 2048);
    hipMemset(tab, 0, sizeof(float) * 2048);
    printf("# %s, %d CUs, 16 waves per CU, %d look-ups per wave\n", p.name, ncu, kIters * kUnroll);
    run<0>("lin", out, tab, ncu);
    run<1>("rnd576", out, tab, ncu);
    run<2>("bperm", out, tab, ncu);
    run<3>("rnd64", out, tab, ncu);
    return 0;
}
