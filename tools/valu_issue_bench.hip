// tools/valu_issue_bench.hip -- gfx950 VALU issue-rate micro-benchmark (measurement tool, not product).
//
// What the demodulator kernels are bound by is VALU issue, so the cost model matters: how many cycles does a
// wave64 take per plain f32 op, per packed-f32 op (v_pk_*_f32), per DPP op, per conversion, per transcendental,
// alone on its SIMD and with 2..4 co-resident waves?  Each test runs ITER iterations of 32 instructions on 8
// independent register chains (or one chain for the latency rows), one wave per workgroup, W workgroups per
// SIMD, and reports cycles per wave-instruction per SIMD from s_memtime (shader clock) and from wall time.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_issue_bench tools/valu_issue_bench.hip && tools/bin/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Op { FMA, PK_FMA, PK_MUL_OPSEL, PK_ADD, ADD_DPP, MOV_DPP, CVT_UBYTE, SQRT, CNDMASK, FMA_DEP, PK_FMA_DEP, MUL, ADD,
          PK_MUL_DEP, FMA_PAIR_DEP, MOV_DPP_WSHL, READLANE, PK_DEP_NOP, PK_IND_NOP, FMA_IND_NOP, RSQ, PERM, CVT_UBYTE1, NOPS };
static const char *kNames[NOPS] = {"v_fma_f32 x8 chains", "v_pk_fma_f32 x8 chains", "v_pk_mul_f32 op_sel x8", "v_pk_add_f32 x8 chains",
                                   "v_add_f32 dpp row_shr:1 x8", "v_mov_b32 dpp row_shr:1 x8", "v_cvt_f32_ubyte0 x8", "v_sqrt_f32 x8",
                                   "v_cndmask_b32 x8", "v_fma_f32 dependent", "v_pk_fma_f32 dependent", "v_mul_f32 x8", "v_add_f32 x8",
                                   "v_pk_mul_f32 dependent", "2 x v_fma_f32 (two chains, dep)", "v_mov_b32 dpp wave_shl:1 x8", "v_readlane_b32 x8",
                                   "v_pk_mul_f32 dependent + s_nop 0", "v_pk_add_f32 x8 + s_nop 0 each", "v_fma_f32 x8 + s_nop 0 each", "v_rsq_f32 x8", "v_perm_b32 x8", "v_cvt_f32_ubyte1 x8"};

template <int OP>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters)
{
    float a[8];
    v2f p[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] * 0.5f}; u[i] = threadIdx.x * 77 + i; }
    const float c = 0.999f, d = 1e-3f;
    const v2f pc{0.999f, 1.001f}, pd{1e-3f, 2e-3f};
    int sacc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            REP32(X)
#undef X
        } else if (OP == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pd));
            REP32(X)
#undef X
        } else if (OP == PK_MUL_OPSEL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(pc));
            REP32(X)
#undef X
        } else if (OP == PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]" : "+v"(p[i]) : "v"(pd));
            REP32(X)
#undef X
        } else if (OP == ADD_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(d));
            REP32(X)
#undef X
        } else if (OP == MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            REP32(X)
#undef X
        } else if (OP == MOV_DPP_WSHL) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            REP32(X)
#undef X
        } else if (OP == CVT_UBYTE) {
#define X(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            REP32(X)
#undef X
        } else if (OP == SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP32(X)
#undef X
        } else if (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
            REP32(X)
#undef X
        } else if (OP == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(d));
            REP32(X)
#undef X
        } else if (OP == PK_FMA_DEP) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pc), "v"(pd));
            REP32(X)
#undef X
        } else if (OP == PK_MUL_DEP) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(p[0]) : "v"(pc));
            REP32(X)
#undef X
        } else if (OP == FMA_PAIR_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 1]) : "v"(c), "v"(d));
            REP32(X)
#undef X
        } else if (OP == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP32(X)
#undef X
        } else if (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
            REP32(X)
#undef X
        } else if (OP == PK_DEP_NOP) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]\n\ts_nop 0" : "+v"(p[0]) : "v"(pc));
            REP32(X)
#undef X
        } else if (OP == PK_IND_NOP) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]\n\ts_nop 0" : "+v"(p[i]) : "v"(pd));
            REP32(X)
#undef X
        } else if (OP == FMA_IND_NOP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(a[i]) : "v"(c), "v"(d));
            REP32(X)
#undef X
        } else if (OP == RSQ) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            REP32(X)
#undef X
        } else if (OP == PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(c), "v"(u[i]), "s"(0x070c0c01));
            REP32(X)
#undef X
        } else if (OP == CVT_UBYTE1) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            REP32(X)
#undef X
        } else if (OP == READLANE) {
#define X(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s_) : "v"(a[i])); sacc ^= s_; }
            REP32(X)
#undef X
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = r + (float)sacc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(int waves_per_simd, int iters, float *d_out, long long *d_cyc, int nsimd)
{
    const int nb = nsimd * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, iters / 8);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(nb), dim3(64), 0, 0, d_out, d_cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(nb);
    hipMemcpy(c.data(), d_cyc, sizeof(long long) * nb, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= nb;
    const double ninst = (double)iters * 32;
    // s_memtime: ticks per instruction as one wave sees it; wall: SIMD-cycles per wave-instruction at 2.4 GHz
    printf("%-34s waves/SIMD %d : memtime ticks/instr/wave %7.2f -> per SIMD %6.2f ; wall %8.3f ms -> %6.2f cyc@2.4GHz per wave-instr per SIMD\n",
           kNames[OP], waves_per_simd, mean / ninst, mean / ninst / waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (ninst * waves_per_simd));
}

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    const int nsimd = pr.multiProcessorCount * 4;
    printf("# %s, %d CUs, clock %d kHz; %d SIMDs; s_memtime tick: see the fma rows (a 64-lane f32 op is 2 or 4 shader cycles)\n",
           pr.name, pr.multiProcessorCount, pr.clockRate, nsimd);
    float *d_out; long long *d_cyc;
    hipMalloc(&d_out, sizeof(float) * 64 * nsimd * 8);
    hipMalloc(&d_cyc, sizeof(long long) * nsimd * 8);
    for (int w : {1, 2, 3, 4}) {
        run<FMA>(w, iters, d_out, d_cyc, nsimd);
        run<MUL>(w, iters, d_out, d_cyc, nsimd);
        run<ADD>(w, iters, d_out, d_cyc, nsimd);
        run<PK_FMA>(w, iters, d_out, d_cyc, nsimd);
        run<PK_MUL_OPSEL>(w, iters, d_out, d_cyc, nsimd);
        run<PK_ADD>(w, iters, d_out, d_cyc, nsimd);
        run<ADD_DPP>(w, iters, d_out, d_cyc, nsimd);
        run<MOV_DPP>(w, iters, d_out, d_cyc, nsimd);
        run<MOV_DPP_WSHL>(w, iters, d_out, d_cyc, nsimd);
        run<CVT_UBYTE>(w, iters, d_out, d_cyc, nsimd);
        run<SQRT>(w, iters, d_out, d_cyc, nsimd);
        run<CNDMASK>(w, iters, d_out, d_cyc, nsimd);
        run<READLANE>(w, iters, d_out, d_cyc, nsimd);
        run<FMA_DEP>(w, iters, d_out, d_cyc, nsimd);
        run<FMA_PAIR_DEP>(w, iters, d_out, d_cyc, nsimd);
        run<PK_FMA_DEP>(w, iters, d_out, d_cyc, nsimd);
        run<PK_MUL_DEP>(w, iters, d_out, d_cyc, nsimd);
        run<PK_DEP_NOP>(w, iters, d_out, d_cyc, nsimd);
        run<PK_IND_NOP>(w, iters, d_out, d_cyc, nsimd);
        run<FMA_IND_NOP>(w, iters, d_out, d_cyc, nsimd);
        run<RSQ>(w, iters, d_out, d_cyc, nsimd);
        run<PERM>(w, iters, d_out, d_cyc, nsimd);
        run<CVT_UBYTE1>(w, iters, d_out, d_cyc, nsimd);
    }
    return 0;
}
