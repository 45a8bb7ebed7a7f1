// tools/cndmask_bench.hip -- what does a v_cndmask_b32 cost on gfx950? profiles/r02_valu_issue.txt shows 23 cycles per wave instruction
// whatever the occupancy; this separates the encodings and operand patterns (VCC vs an SGPR pair, in-place vs fresh destination,
// alternated with the compare that feeds it) so that the kernels can avoid the slow form -- if it is real and not an artefact.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/cndmask_bench tools/cndmask_bench.hip && tools/bin/cndmask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
enum { CND_VCC_INPLACE, CND_VCC_FRESH, CND_SGPR, CMP_CND, MAXF, CND_SRC0_CONST, AND_OR, CMP_4CND, CMPS_4CND, CMP_4CND_MIX, NOPS };
static const char *kNames[NOPS] = {"v_cndmask_b32 a,a,c,vcc (in place)", "v_cndmask_b32 b,a,c,vcc (other destination)", "v_cndmask_b32 b,a,c,s[mask]",
                                   "v_cmp_gt_f32 + v_cndmask_b32 pair", "v_max_f32 (reference)", "v_cndmask_b32 b,0,a,vcc", "v_and_b32 + v_or_b32 select (2 instr)", "v_cmp -> vcc, then 4 v_cndmask vcc (5 instr)", "v_cmp -> s[], then 4 v_cndmask s[] (5 instr)",
                                   "v_cmp -> vcc, 4 x (v_cndmask vcc + v_max_f32) (9 instr)"};

template <int OP>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters)
{
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; b[i] = 0.f; }
    const float c = 0.999f;
    unsigned long long m = (threadIdx.x & 1) ? 0xaaaaaaaaaaaaaaaaull : 0x5555555555555555ull;
    m = __builtin_amdgcn_read_exec() & 0x5555555555555555ull;
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(1.5f) : "vcc");
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == CND_VCC_INPLACE) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            REP32(X)
#undef X
        } else if (OP == CND_VCC_FRESH) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(b[i]) : "v"(a[i]), "v"(c));
            REP32(X)
#undef X
        } else if (OP == CND_SGPR) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(b[i]) : "v"(a[i]), "v"(c), "s"(m));
            REP32(X)
#undef X
        } else if (OP == CMP_CND) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(b[i]) : "v"(a[i]), "v"(c) : "vcc");
            REP32(X)
#undef X
        } else if (OP == MAXF) {
#define X(i) asm volatile("v_max_f32 %0, %1, %2" : "=v"(b[i]) : "v"(a[i]), "v"(c));
            REP32(X)
#undef X
        } else if (OP == CND_SRC0_CONST) {
#define X(i) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(b[i]) : "v"(a[i]));
            REP32(X)
#undef X
        } else if (OP == CMP_4CND) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\tv_cndmask_b32 %2, %4, %5, vcc\n\tv_cndmask_b32 %3, %5, %4, vcc" \
                          : "=&v"(b[i]), "=&v"(b[(i + 1) & 7]), "=&v"(b[(i + 2) & 7]), "=&v"(b[(i + 3) & 7]) : "v"(a[i]), "v"(c) : "vcc");
            REP32(X)
#undef X
        } else if (OP == CMPS_4CND) {
#define X(i) { unsigned long long m_; asm volatile("v_cmp_gt_f32 %4, %5, %6\n\tv_cndmask_b32 %0, %5, %6, %4\n\tv_cndmask_b32 %1, %6, %5, %4\n\tv_cndmask_b32 %2, %5, %6, %4\n\tv_cndmask_b32 %3, %6, %5, %4" \
                          : "=&v"(b[i]), "=&v"(b[(i + 1) & 7]), "=&v"(b[(i + 2) & 7]), "=&v"(b[(i + 3) & 7]), "=&s"(m_) : "v"(a[i]), "v"(c)); }
            REP32(X)
#undef X
        } else if (OP == CMP_4CND_MIX) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_max_f32 %4, %4, %5\n\tv_cndmask_b32 %1, %5, %4, vcc\n\tv_max_f32 %4, %4, %5\n\tv_cndmask_b32 %2, %4, %5, vcc\n\tv_max_f32 %4, %4, %5\n\tv_cndmask_b32 %3, %5, %4, vcc\n\tv_max_f32 %4, %4, %5" \
                          : "=&v"(b[i]), "=&v"(b[(i + 1) & 7]), "=&v"(b[(i + 2) & 7]), "=&v"(b[(i + 3) & 7]), "+v"(a[i]) : "v"(c) : "vcc");
            REP32(X)
#undef X
        } else if (OP == AND_OR) {
#define X(i) asm volatile("v_and_b32 %0, %1, %2\n\tv_or_b32 %0, %0, %3" : "=&v"(b[i]) : "v"(a[i]), "v"(c), "v"(a[(i + 1) & 7]));
            REP32(X)
#undef X
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + b[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(int wps, int iters, float *d_out, long long *d_cyc, int nsimd)
{
    const int nb = nsimd * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, iters / 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ninst = (double)iters * 32;
    printf("%-46s waves/SIMD %d : wall %8.3f ms -> %6.2f cyc@2.4GHz per asm statement per SIMD\n", kNames[OP], wps, ms, ms * 1e-3 * 2.4e9 / (ninst * wps));
}

int main()
{
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    const int nsimd = pr.multiProcessorCount * 4, iters = 20000;
    float *d_out; long long *d_cyc;
    hipMalloc(&d_out, sizeof(float) * 64 * nsimd * 8);
    hipMalloc(&d_cyc, sizeof(long long) * nsimd * 8);
    printf("# %s: cost of v_cndmask_b32 forms (tools/cndmask_bench.hip)\n", pr.name);
    for (int w : {1, 3}) {
        run<MAXF>(w, iters, d_out, d_cyc, nsimd);
        run<CND_VCC_INPLACE>(w, iters, d_out, d_cyc, nsimd);
        run<CND_VCC_FRESH>(w, iters, d_out, d_cyc, nsimd);
        run<CND_SGPR>(w, iters, d_out, d_cyc, nsimd);
        run<CND_SRC0_CONST>(w, iters, d_out, d_cyc, nsimd);
        run<CMP_CND>(w, iters, d_out, d_cyc, nsimd);
        run<AND_OR>(w, iters, d_out, d_cyc, nsimd);
        run<CMP_4CND>(w, iters, d_out, d_cyc, nsimd);
        run<CMPS_4CND>(w, iters, d_out, d_cyc, nsimd);
        run<CMP_4CND_MIX>(w, iters, d_out, d_cyc, nsimd);
    }
    return 0;
}
