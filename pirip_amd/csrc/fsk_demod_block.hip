// pirip_amd/csrc/fsk_demod_block.hip -- workgroup-per-stream FSK demodulator for LONG symbols on gfx950 (MI355X).
//
// `rtl_fsk -r 1000` at the default 240 kS/s (/root/reference/README.md:152,184; `-m 4 --mask 2000`: README.md:239) means Ts = 240 samples
// per symbol, 12 000-sample frames and, by fsk_create_core's rule, a 4096-point estimator FFT. A wave-per-stream instance would need
// ~85 KB of LDS per wave (one wave per CU); the any-configuration kernel runs this shape with run-time loops at 45 G samples/s
// (profiles/r03_instance_rates.txt, 223 SGPR spills). This file is the shape as template arguments, one 256-thread workgroup per stream:
//   * estimator FFT: 16 points per thread, the six kiss_fft radix-4 stages as three register passes of two stages each with two
//     exchanges through one 34 KB LDS array -- the same butterfly network on the same operands with the same twiddles as kiss_fft
//     [UPSTREAM-RECALLED kiss_fft.c kf_bfly4], so Sf / f_est stay bit-identical; which thread computes which
//     butterfly is bookkeeping:
//       input index i = e0 + 4 e1 + 16 e2 + 64 e3 + 256 e4 + 1024 e5 (base-4 digits); stage s is a 4-point DFT over digit e(6-s)
//       pass A  thread (e3 e2 e1 e0) = t       holds inputs t + 256 n, n = e4 + 4 e5 (coalesced reads); stages m = 1, 4
//                                              -> R[K], K = k0 + 4 k1
//       pass B  thread (K, e1 e0)              collects R[K] over (e3, e2); stages m = 16, 64 with k = K and K + 16 k2
//                                              -> V[k3 + 4 k2] = slot K2 = K + 16 k2 + 64 k3
//       pass C  thread K2                      collects over (e1, e0); stages m = 256, 1024 with k = K2 and K2 + 256 k4
//                                              -> bins K2 + 256 k4 + 1024 k5: a thread ends up with 16 bins that are 256 apart, the
//                                              fftshift keeps them with it, so **Sf lives in 16 registers per thread**
//     exchange layouts: A->B xa[K * 272 + t], B->C xa[K2 * 17 + e10]: every wave-wide 8-byte access touches each LDS bank exactly twice
//     (the minimum) on both sides;
//   * correlator: every thread owns a run of three consecutive 16-sample window steps (twelve threads: four), samples read from global
//     memory (L2) as 16-byte pieces and converted once for two tones, per-thread restart of the upstream oscillator recursion (table phasor x
//     first-order gain drift, then codec2's float32-rounded per-sample multiplier), sums over the 16-sample steps into LDS; no f_dc
//     memory: the last frame's 540 raw samples are kept (LDS, 1 KB) and mixed again with last frame's tone estimates, both oscillators
//     at the phase reference between the last old and the first new sample (see fsk_demod_wave.hip, round 4);
//   * window sums = 15 consecutive step sums, fine timing, decisions as in the other kernels.
// Numerics as DESIGN.md 5: Sf, f_est bit-exact; nin exact away from threshold ties; rx_filt within tolerance (scaled by N/2400).
// The argument block is copied to LDS once and reached through a pointer that is made opaque at every phase: by value it was live
// across the frame loop in the general kernel and cost 223 SGPR spills.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/pirip_hip.h"
#include "fsk_device.hpp"

namespace pirip {

namespace {

constexpr int kWave = 64;
constexpr int NT = 256;                 // threads per stream
constexpr int TS = 240, NSYM = 50, P = 15, NDFT = 4096, LOG2N = 12;
constexpr int N = TS * NSYM, Q = TS / 4, NMEM = N + 2 * TS, HIST = 2 * TS + Q, STEP = TS / P, NINT = (NSYM + 1) * P;
constexpr int NFFT = (N - Q) / (NDFT / 2) - 1;
constexpr int NSTEP = NMEM / STEP;      // 780 sixteen-sample steps of integrator memory
// correlator: every thread owns a run of consecutive steps -- three each, and the 12 steps left over go to the first 12 threads of the last
// wave (four steps there): 13 wave-passes over a step instead of the 16 that 195 threads x 4 steps took
constexpr int CSTEPS = NSTEP / NT;      // 3
constexpr int CEXTRA = NSTEP - CSTEPS * NT;   // 12
static_assert(NFFT == 4 && (N + Q) / (NDFT / 2) - 1 == NFFT, "four FFTs per frame whatever nin");
static_assert(NSTEP * STEP == NMEM && CSTEPS * NT <= NSTEP && CEXTRA >= 0 && CEXTRA <= kWave, "integrator memory divides into steps; the left-over steps fit one wave");
static_assert(NINT + P - 1 <= NSTEP, "the last window ends inside the memory");
constexpr int XA_CF = 16 * 272;         // exchange array, complex floats (34 816 bytes)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));      // 16 bytes at the alignment of an (I, Q) byte pair

// ---- packed-f32 complex helpers (as in fsk_demod_wave.hip: kiss_fft's arithmetic, every product and sum rounded once) ----------
__device__ __forceinline__ v2f cmul_x(v2f a, v2f t)
{
    v2f p2, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]"
        : "=&v"(r), "=&v"(p2) : "v"(a), "v"(t));
    return r;
}
__device__ __forceinline__ v2f add_rot(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f sub_rot(v2f a, v2f b) { v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void bfly4(v2f &f0, v2f &f1, v2f &f2, v2f &f3)
{
    const v2f s5 = f0 - f2;
    f0 = f0 + f2;
    const v2f s3 = f1 + f3;
    const v2f s4 = f1 - f3;
    f2 = f0 - s3;
    f0 = f0 + s3;
    f1 = add_rot(s5, s4);
    f3 = sub_rot(s5, s4);
}
// two radix-4 stages over 16 register values X[c + 4 dd]: first over dd for every c (twiddles t1[0..2], the same for every c; nullptr:
// trivial), results at X[c + 4 k]; then over c for every k with twiddles t2[3 k + 0..2], results k' at X[k' + 4 k]
__device__ __forceinline__ void radix16(v2f *X, const v2f *t1, const v2f *t2, bool first_trivial)
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v2f f0 = X[c], f1 = X[c + 4], f2 = X[c + 8], f3 = X[c + 12];
        if (!first_trivial) { f1 = cmul_x(f1, t1[0]); f2 = cmul_x(f2, t1[1]); f3 = cmul_x(f3, t1[2]); }
        bfly4(f0, f1, f2, f3);
        X[c] = f0; X[c + 4] = f1; X[c + 8] = f2; X[c + 12] = f3;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v2f f0 = X[4 * k], f1 = X[4 * k + 1], f2 = X[4 * k + 2], f3 = X[4 * k + 3];
        if (!(first_trivial && k == 0)) { f1 = cmul_x(f1, t2[3 * k]); f2 = cmul_x(f2, t2[3 * k + 1]); f3 = cmul_x(f3, t2[3 * k + 2]); }
        bfly4(f0, f1, f2, f3);
        X[4 * k] = f0; X[4 * k + 1] = f1; X[4 * k + 2] = f2; X[4 * k + 3] = f3;
    }
}
__device__ __forceinline__ v2f mix_conj(v2f x, v2f ph)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(x), "v"(ph));
    return r;
}
// acc + x * conj(ph): the down-conversion folded into the running sum, 2 packed fma (round 5)
__device__ __forceinline__ v2f mix_conj_acc(v2f x, v2f ph, v2f acc)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(x), "v"(ph), "v"(acc));
    return r;
}
__device__ __forceinline__ v2f rot_step(v2f ph, v2f d)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(ph), "v"(d));
    return r;
}
// correctly rounded sqrt for x = 0 or x >= 2^-96 (fsk_demod_wave.hip: measured on the device over every float; pirip_hip_selftest_sqrt)
// NZ: every value of the batch >= 2^-96 (none zero): the clamp is a no-op there and is left out
template <bool NZ = false>
__device__ __forceinline__ float sqrt_rn_fast(float x)
{
    float q = __builtin_amdgcn_rsqf(x);
    if (!NZ) asm("v_min_f32 %0, %0, %1" : "+v"(q) : "v"(0x1p60f));
    const float y = x * q, h = 0.5f * q;
    return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
}
template <int FMT>
__device__ __forceinline__ float cvt_u8(float b)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD) return __builtin_fmaf(b, 0.0078125f, -0.9921875f);
    return __builtin_fmaf(b, -1.187418e-07f, __builtin_fmaf(b, 0.007843255996704102f, -1.0f));
}
// the same maps on an (I, Q) pair of byte values: one v_pk_fma_f32 per fma (round 5, as in fsk_demod_wave.hip)
template <int FMT>
__device__ __forceinline__ v2f cvt_u8_pair(v2f b)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD) return __builtin_elementwise_fma(b, v2f{0.0078125f, 0.0078125f}, v2f{-0.9921875f, -0.9921875f});
    return __builtin_elementwise_fma(b, v2f{-1.187418e-07f, -1.187418e-07f},
                                     __builtin_elementwise_fma(b, v2f{0.007843255996704102f, 0.007843255996704102f}, v2f{-1.0f, -1.0f}));
}
template <int FMT>
__device__ __forceinline__ v2f cvt_sample_hi(uint32_t v)  // high 16 bits: (I, Q) bytes
{
    return cvt_u8_pair<FMT>(v2f{(float)((v >> 16) & 0xffu), (float)(v >> 24)});
}
template <int FMT>
__device__ __forceinline__ v2f cvt_sample(uint32_t v)     // low 16 bits: (I, Q) bytes
{
    return cvt_u8_pair<FMT>(v2f{(float)(v & 0xffu), (float)((v >> 8) & 0xffu)});
}

// The argument block is read from LDS, i.e. into vector registers; pointers and sizes in it are wave-uniform all the same. Moved to
// scalar registers, global accesses through them use the scalar-base + 32-bit-offset form instead of a 64-bit address pair per access.
template <class T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// ... and they point into device memory: read out of the LDS copy of the argument block the compiler only knows them as generic pointers
// and every access through them is a FLAT instruction (a 64-bit address per lane, an LDS-aperture check, and a slot in the LDS wait counter
// as well as the memory one: 196 of them in the first cut). gl() says "global", which gives scalar-base + 32-bit-offset global accesses.
#define PIRIP_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ PIRIP_GLOBAL T *gl(T *p) { return (PIRIP_GLOBAL T *)uni(p); }
template <class T>
__device__ __forceinline__ PIRIP_GLOBAL T *gl(PIRIP_GLOBAL T *p)     // a global pointer advanced by a wave-uniform amount: back to scalar registers
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (PIRIP_GLOBAL T *)(((unsigned long long)hi << 32) | lo);
}
// element idx of a global array. (Measured, interleaved A/B on one box: forming the byte offset in 32 bits -- scalar base + 32-bit lane
// offset for EVERY access of the kernel -- ran 13 % slower than plain indexing, 205 against 235 G samples/s, with fewer instructions; the
// row-base form below is used where the row is wave-uniform and kept because it measured faster there. -DPIRIP_BLOCK_LDG32 selects it.)
template <class T>
__device__ __forceinline__ T ldg(const PIRIP_GLOBAL T *base, unsigned idx)
{
#ifdef PIRIP_BLOCK_LDG32
    return *(const PIRIP_GLOBAL T *)((const PIRIP_GLOBAL char *)base + idx * (unsigned)sizeof(T));
#else
    return base[(int)idx];
#endif
}
// the same with a wave-uniform row base: base and row advance in scalar registers, the lane's own offset is the only vector operand
template <class T>
__device__ __forceinline__ T ldrow(const PIRIP_GLOBAL T *row_base, unsigned lane_elem) { return *(const PIRIP_GLOBAL T *)((const PIRIP_GLOBAL char *)row_base + lane_elem * (unsigned)sizeof(T)); }

#define PIRIP_DPP_F(old, src, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_sum(float v)
{
    v += PIRIP_DPP_F(0.f, v, 0x111, 0xf);
    v += PIRIP_DPP_F(0.f, v, 0x112, 0xf);
    v += PIRIP_DPP_F(0.f, v, 0x114, 0xf);
    v += PIRIP_DPP_F(0.f, v, 0x118, 0xf);
    v += PIRIP_DPP_F(0.f, v, 0x142, 0xa);
    v += PIRIP_DPP_F(0.f, v, 0x143, 0xc);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load in flight (vmcnt(0)), which would
// put the latency of the twiddle loads issued in front of an exchange back on the critical path.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// block reductions over the 4 waves; every thread gets the result. red: 16 words of LDS.
__device__ __forceinline__ float block_sum(float v, float *red, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
// two sums at once (one barrier pair)
__device__ __forceinline__ void block_sum2(float &a, float &b, float *red, int tid)
{
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = a; red[4 + (tid >> 6)] = b; }
    __syncthreads();
    a = ((red[0] + red[1]) + red[2]) + red[3];
    b = ((red[4] + red[5]) + red[6]) + red[7];
}
// arg-max with codec2's tie rule (first maximum wins): larger value, then smaller index
__device__ __forceinline__ void block_argmax(float &v, int &idx, float *red, int tid)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const float ov = __shfl_xor(v, s);
        const int oi = __shfl_xor(idx, s);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = v; ((int *)red)[8 + (tid >> 6)] = idx; }
    __syncthreads();
    v = red[0]; idx = ((int *)red)[8];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        const float ov = red[w]; const int oi = ((int *)red)[8 + w];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

}  // namespace

// timing experiments only (results wrong): fewer FFTs / no correlator threads -- what a phase costs is the time its removal saves
#ifndef PIRIP_BLOCK_T_NFFT
#define PIRIP_BLOCK_T_NFFT NFFT
#endif
#ifndef PIRIP_BLOCK_T_NCORR
#define PIRIP_BLOCK_T_NCORR NT
#endif
#ifndef PIRIP_BLOCK_PREFETCH
#define PIRIP_BLOCK_PREFETCH 1
#endif
#ifndef PIRIP_BLOCK_WPB2
#define PIRIP_BLOCK_WPB2 3     // workgroups per CU the 2-FSK instances are compiled for (36 KB of LDS each, <= 168 VGPR)
#endif
#ifndef PIRIP_BLOCK_WPB4
#define PIRIP_BLOCK_WPB4 3     // ... and the 4-FSK instances (51 KB of LDS; tones two at a time in the correlator). Round 5: with the down-conversion
                               // folded into the running sums the peak instances need 156 VGPR (were 216) and run at three as well
#endif
template <int M, int FMT, bool MASK>
__global__ __launch_bounds__(NT, M == 2 ? PIRIP_BLOCK_WPB2 : PIRIP_BLOCK_WPB4) void fsk_demod_block_kernel(DemodArgs a_by_value)
{
    // The argument block is copied to LDS once; every phase re-derives what it needs through a pointer that is made opaque per phase, so
    // nothing of the block stays live in registers across the frame loop (by value it cost the general kernel 223 SGPR spills).
    __shared__ DemodArgs s_args;
    if (threadIdx.x == 0) s_args = a_by_value;
    __syncthreads();
    const DemodArgs *ap = &s_args;
#define PIRIP_ARGS() asm volatile("" : "+v"(ap)); const DemodArgs &a = *ap; (void)a

    // one work array: the FFT exchanges (XA_CF), then the linear spectrum (mask estimator), then the sums over the 16-sample window
    // steps [M][NSTEP] with f_int [M][NINT] behind them
    constexpr int WORK_CF = XA_CF > M * (NSTEP + NINT) ? XA_CF : M * (NSTEP + NINT);
    __shared__ __attribute__((aligned(16))) float2 s_xa[WORK_CF];
    float2 (*s_step)[NSTEP] = (float2 (*)[NSTEP])s_xa;
    __shared__ __attribute__((aligned(16))) uint16_t s_tail[HIST + 4];      // last frame's raw tail (I, Q bytes per sample)
    __shared__ float s_red[16];
#if PIRIP_BLOCK_PREFETCH
    __shared__ uint32_t s_dump[kWave];      // where the line-touching prefetch of the next frame lands (a wave writes lane-linear: 64 dwords)
#endif
    __shared__ uint32_t s_prev[2 * kMaxTones + 1];   // last frame's phase steps [M], oscillator-table rows [M], nin: read once per frame by the correlator
    __shared__ float s_sc[6];               // SNRest, snr_est, EbNodB, v_est, rx_sig_pow, rx_nse_pow: stream state that only wave 0 touches, on observable frames
    __shared__ float s_fest[kMaxTones];     // the latest frame's tone estimates (thread 0 writes them, and reads them back when the state is saved)
    __shared__ __attribute__((aligned(16))) float2 s_twb[16 * 16];         // pass-B twiddles of the FFT: [K][15], they depend on K = tid >> 4 only

    const int tid = threadIdx.x;
    const int sid = blockIdx.x;
    int64_t pos = 0;
    int frame = 0;
    int nin;
    float SfR[16];
    float sc_norm_rx_timing, sc_ppm;
    bool have_frames = false;
    {
        PIRIP_ARGS();
        if (a.io.seg && a.io.seg[sid].max_frames < 0) return;
        const StreamScalars sc = a.s.scal[sid];
        nin = sc.nin; sc_norm_rx_timing = sc.norm_rx_timing; sc_ppm = sc.ppm;
        if (tid == 0) { s_sc[0] = sc.SNRest; s_sc[1] = sc.snr_est; s_sc[2] = sc.EbNodB; s_sc[3] = sc.v_est; s_sc[4] = sc.rx_sig_pow; s_sc[5] = sc.rx_nse_pow; }
        // owned bins K2 + 256 k4 + 1024 k5 at register k5 + 4 k4; Sf is stored fftshifted: index (bin + Ndft/2) mod Ndft
#pragma unroll
        for (int r = 0; r < 16; r++) SfR[r] = gl(a.s.Sf + (size_t)sid * NDFT)[(tid + 256 * (r >> 2) + 1024 * (r & 3) + NDFT / 2) & (NDFT - 1)];
        // state block of this stream: raw tail (HIST x 2 bytes), then per tone last frame's phase step and table row, then its nin
        const PIRIP_GLOBAL uint32_t *st32 = gl((const uint32_t *)(a.s.hist + (size_t)sid * M * HIST));
        constexpr int TR = (HIST * 2 + 15) / 16 * 4;       // first trailer dword
        if (tid == 0) s_prev[2 * M] = st32[TR + 2 * M];
#pragma unroll
        for (int m = 0; m < M; m++) if (tid == 0) { s_prev[m] = st32[TR + m]; s_prev[M + m] = st32[TR + M + m]; }
        for (int i = tid; i < HIST / 2; i += NT) ((uint32_t *)s_tail)[i] = st32[i];
        {   // slot i of row K: i < 3: tw[K 64 r], r = i + 1 (stage m = 16); else tw[(K + 16 k2) 16 r], k2 = (i - 3) / 3, r = (i - 3) % 3 + 1 (m = 64)
            const int K = tid >> 4, i = tid & 15;
            const int k2 = i < 3 ? 0 : (i - 3) / 3, r = i < 3 ? i + 1 : (i - 3) % 3 + 1;
            const unsigned idx = i < 3 ? (unsigned)(K * 64 * r) : (unsigned)((K + 16 * k2) * 16 * r);
            s_twb[tid] = a.t.tw[i < 15 ? idx : 0];
        }
    }
    __syncthreads();

    int max_frames;
    int64_t nsamp;
    const PIRIP_GLOBAL uint8_t *in_base;
    int64_t out0 = 0;
    {
        PIRIP_ARGS();
        max_frames = (int)(a.io.max_frames < 0x7fffffff ? a.io.max_frames : 0x7fffffff); nsamp = a.io.nsamp;
        in_base = gl(a.io.in + (size_t)sid * a.io.in_stride);
        if (a.io.seg) { const SegDesc sd = a.io.seg[sid]; in_base += (size_t)sd.in_off * 2; nsamp -= sd.in_off; out0 = sd.out_frame0; max_frames = sd.max_frames; }
    }

    {
        // Workgroups that start together on a CU walk their frames in lockstep: all of them are in the FFT passes at once and all of them in
        // the barrier-separated small phases at once. An experiment (PIRIP_BLOCK_STAGGER=<units>, 0 = off): each wave waits wave-slot x units
        // x 8128 clocks before its first frame (HW_ID.wave_id: the workgroups of a CU's first round sit in slots 0, 1, 2 of every SIMD).
        PIRIP_ARGS();
        const int stg = a.d.block_stagger;
        if (stg > 0) {
            const unsigned slot = __builtin_amdgcn_s_getreg(0x1804) % 3u;
            for (unsigned i = 0; i < slot * (unsigned)stg; i++) __builtin_amdgcn_s_sleep(127);
        }
    }
    while (frame < max_frames && pos + nin <= nsamp) {
        const PIRIP_GLOBAL uint16_t *gin = gl((const PIRIP_GLOBAL uint16_t *)(in_base + 2 * pos));   // this frame's samples (I, Q bytes)
        const int nold = NMEM - nin;
        // ================= a-5: frequency estimator =====================================================================
        {
            PIRIP_ARGS();
            const PIRIP_GLOBAL v2f *__restrict__ g_tw = gl((const v2f *)a.t.tw);
            const PIRIP_GLOBAL float *__restrict__ g_hann = gl(a.t.hann);
            const PIRIP_GLOBAL v2f *__restrict__ g_twc = gl((const v2f *)a.t.fast_tab);      // last pass's twiddles, [15][256] (fsk_plan.cpp)
            const float k1mtc = a.d.one_minus_tc, ktc = a.d.tc;
            // twiddles of this thread's butterflies: pass A wave-uniform (scalar loads); passes B and C per thread, fetched from the
            // 32 KB table (L1 / L2) right before each pass -- 30 loads per FFT against 78 registers held for the whole frame
            auto TW = [&](unsigned idx) { return ldg(g_tw, idx); };
            // (no software prefetch and no register-resident Hann samples or twiddles: each is fetched from L1 / L2 where it is used. With
            //  prefetches the kernel held 252 VGPR = two workgroups per CU and ran 193 G samples/s (2-FSK); at <= 168 VGPR a third
            //  workgroup fits and its waves hide the same latencies: 228 G)
#pragma unroll 1
            for (int j = 0; j < PIRIP_BLOCK_T_NFFT; j++) {
                // (an opaque copy of the thread index per FFT: the twiddle loads below are loop-invariant, and hoisted out of this
                //  loop they are 54 more live registers for the whole frame)
                int tq = tid; asm volatile("" : "+v"(tq));
                const int Kb = tq >> 4, e10 = tq & 15;
                v2f X[16];
                {
                    const PIRIP_GLOBAL uint16_t *srow = gin + (NDFT / 2) * j;       // wave-uniform rows of 256 samples / 256 window values
#pragma unroll
                    for (int n = 0; n < 16; n++) {         // n = e4 + 4 e5 -> X[c + 4 dd], c = e4, dd = e5
                        const v2f x = cvt_sample<FMT>((uint32_t)ldrow(srow + 256 * n, (unsigned)tq));
                        const float hn = ldrow(g_hann + 256 * n, (unsigned)tq);
                        X[n] = v2f{hn * x.x, hn * x.y};
                    }
                }
                // pass A: m = 1 (trivial twiddles) over e5, then m = 4 over e4 with tw[256 k0 r]
                {
                    v2f t2[12];
#pragma unroll
                    for (int i = 0; i < 3; i++) t2[i] = v2f{1.f, 0.f};
#pragma unroll
                    for (int k = 1; k < 4; k++)
#pragma unroll
                        for (int r = 1; r < 4; r++) t2[3 * k + (r - 1)] = TW(256u * k * r);                 // m = 4: tw[k fs r], fs = 256
                    radix16(X, nullptr, t2, true);
                }
                // X[k1 + 4 k0] now; R[K] with K = k0 + 4 k1 is X[(K >> 2) + 4 (K & 3)]
                lds_barrier();                             // the previous FFT's (or frame's) reads of the array are done
#pragma unroll
                for (int K = 0; K < 16; K++) { const v2f v = X[(K >> 2) + 4 * (K & 3)]; s_xa[K * 272 + tid] = make_float2(v.x, v.y); }
                // pass B: m = 16 over e3 (k = K, fs = 64), then m = 64 over e2 (k = K + 16 k2, fs = 16); its 15 twiddles depend on K only and
                // sit in LDS (2 KB for the workgroup: immediate-offset reads instead of 15 global loads with an address each)
                {
                    v2f tb1[3], tb2[12];
                    lds_barrier();
                    {
                        const float2 *twb = s_twb + Kb * 16;
#pragma unroll
                        for (int i = 0; i < 3; i++) { const float2 w = twb[i]; tb1[i] = v2f{w.x, w.y}; }
#pragma unroll
                        for (int i = 0; i < 12; i++) { const float2 w = twb[3 + i]; tb2[i] = v2f{w.x, w.y}; }
                    }
#pragma unroll
                    for (int e = 0; e < 16; e++) { const float2 v = s_xa[Kb * 272 + e * 16 + e10]; X[e] = v2f{v.x, v.y}; }   // e = e2 + 4 e3
                    radix16(X, tb1, tb2, false);
                }
                // X[k3 + 4 k2] is slot K2 = Kb + 16 k2 + 64 k3
                lds_barrier();
#pragma unroll
                for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
                    for (int k3 = 0; k3 < 4; k3++) { const v2f v = X[k3 + 4 * k2]; s_xa[(Kb + 16 * k2 + 64 * k3) * 17 + e10] = make_float2(v.x, v.y); }
                // pass C: m = 256 over e1 (k = K2, fs = 4), then m = 1024 over e0 (k = K2 + 256 k4, fs = 1); twiddles requested likewise
                {
                    v2f tc1[3], tc2[12];
#pragma unroll
                    for (int i = 0; i < 3; i++) tc1[i] = ldrow(g_twc + 256 * i, (unsigned)tq);
#pragma unroll
                    for (int i = 0; i < 12; i++) tc2[i] = ldrow(g_twc + 256 * (3 + i), (unsigned)tq);
                    lds_barrier();
#pragma unroll
                    for (int e = 0; e < 16; e++) { const float2 v = s_xa[tid * 17 + e]; X[e] = v2f{v.x, v.y}; }                // e = e0 + 4 e1
                    radix16(X, tc1, tc2, false);
                }
                // X[k5 + 4 k4] = bin K2 + 256 k4 + 1024 k5: |X|, smoothing (this thread owns these bins)
                float mg[16];
                unsigned kmin = 0xffffffffu;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    mg[r] = (X[r].x * X[r].x) + (X[r].y * X[r].y);
                    const unsigned key = __builtin_bit_cast(unsigned, mg[r]);       // (non-negative floats order as unsigned integers)
                    kmin = key < kmin ? key : kmin;
                }
                if (__all(kmin >= 0x0f800000u)) {                                   // every |X|^2 >= 2^-96: roots without the zero guard
#pragma unroll
                    for (int r = 0; r < 16; r++) mg[r] = sqrt_rn_fast<true>(mg[r]);
                } else {
                    kmin = 0xffffffffu;
#pragma unroll
                    for (int r = 0; r < 16; r++) { const unsigned key = __builtin_bit_cast(unsigned, mg[r]) - 1u; kmin = key < kmin ? key : kmin; }
                    if (__all(kmin >= 0x0f800000u - 1u)) {                          // zeros among them (a silent input)
#pragma unroll
                        for (int r = 0; r < 16; r++) mg[r] = sqrt_rn_fast(mg[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++) mg[r] = sqrtf(mg[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) SfR[r] = (SfR[r] * k1mtc) + (mg[r] * ktc);
            }
        }
        // ---- tone estimate ---------------------------------------------------------------------------------------------
        int freqi[M];
        int bb = 0;
        uint32_t dthv[M];
        int tix[M];
        float f_est[kMaxTones] = {0.f, 0.f, 0.f, 0.f};
        {
            PIRIP_ARGS();
            const FskDims &d = a.d;
            int sfi[16];
#pragma unroll
            for (int r = 0; r < 16; r++) sfi[r] = (tid + 256 * (r >> 2) + 1024 * (r & 3) + NDFT / 2) & (NDFT - 1);
            if constexpr (MASK) {
                float *sfl = (float *)s_xa;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; r++) sfl[sfi[r]] = SfR[r];
                __syncthreads();
                const int est_st = uni(d.est_st), b_end = uni(d.est_en - d.mask_len), n_teeth = uni(d.n_teeth);
                const PIRIP_GLOBAL int16_t *__restrict__ g_teeth = gl(a.t.teeth);
                float best = 0.0f; int ib = est_st;
#pragma unroll 1
                for (int b = est_st + tid; b < b_end; b += NT) {
                    float corr = 0.0f;
                    for (int k = 0; k < n_teeth; k++) corr += sfl[b + g_teeth[k]];
                    if (corr > best) { best = corr; ib = b; }
                }
                block_argmax(best, ib, s_red, tid);
                bb = ib;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    freqi[m] = 0;
                    f_est[m] = (float)((bb - NDFT / 2) * d.Fs / NDFT) + (float)(m * d.tone_spacing);
                    dthv[m] = gl(a.t.mask_dtheta)[bb * M + m]; tix[m] = bb * M + m;
                }
            } else {
                float w[16];
                const int est_st = uni(d.est_st), est_en = uni(d.est_en), f_zero = uni(d.f_zero);
#pragma unroll
                for (int r = 0; r < 16; r++) w[r] = SfR[r];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    float best = 0.0f; int ib = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (sfi[r] >= est_st && sfi[r] < est_en && (w[r] > best || (w[r] == best && best > 0.0f && sfi[r] < ib))) { best = w[r]; ib = sfi[r]; }
                    block_argmax(best, ib, s_red, tid);
                    int f_min = ib - f_zero; f_min = f_min < 0 ? 0 : f_min;
                    int f_max = ib + f_zero; f_max = f_max > NDFT ? NDFT : f_max;
#pragma unroll
                    for (int r = 0; r < 16; r++) if (sfi[r] >= f_min && sfi[r] < f_max) w[r] = 0.0f;
                    freqi[m] = ib - NDFT / 2;
                }
#pragma unroll
                for (int x = 1; x < M; x++)
#pragma unroll
                    for (int y = x; y > 0; y--)
                        if (freqi[y] < freqi[y - 1]) { const int t = freqi[y]; freqi[y] = freqi[y - 1]; freqi[y - 1] = t; }
#pragma unroll
                for (int m = 0; m < M; m++) {
                    f_est[m] = (float)freqi[m] * d.bin_hz;
                    dthv[m] = (uint32_t)freqi[m] << (32 - LOG2N); tix[m] = freqi[m] + NDFT / 2;
                }
            }
        }
        // ================= a-6: down-convert, sums over the 16-sample window steps ========================================
        __syncthreads();                                   // (the linear spectrum in s_xa has been read)
        if (tid < PIRIP_BLOCK_T_NCORR) {
            PIRIP_ARGS();
            const PIRIP_GLOBAL v2f *__restrict__ g_tw = gl((const v2f *)a.t.tw);
            const PIRIP_GLOBAL v2f *__restrict__ g_step = gl((const v2f *)a.t.osc_step), *__restrict__ g_drift = gl((const v2f *)a.t.osc_drift);
            auto phasor = [&](uint32_t th) {
                const v2f w = ldg(g_tw, th >> (32 - LOG2N));
                float pc = w.x, ps = -w.y;
                if constexpr (MASK) {
                    const float bl = (float)(th & ((1u << (32 - LOG2N)) - 1u)) * 1.4629180792671596e-9f;
                    const float b2 = bl * bl;
                    const float cb = 1.0f - b2 * (0.5f - b2 * (1.0f / 24.0f));
                    const float sb = bl * (1.0f - b2 * ((1.0f / 6.0f) - b2 * (1.0f / 120.0f)));
                    const float c2 = pc * cb - ps * sb, s2 = ps * cb + pc * sb;
                    pc = c2; ps = s2;
                }
                return v2f{pc, ps};
            };
            const int xt = tid - (NT - kWave);             // lane of the last wave
            const int step0 = CSTEPS * tid + (xt < 0 ? 0 : xt < CEXTRA ? xt : CEXTRA);     // this thread's first step
            const int nsteps = CSTEPS + (xt >= 0 && xt < CEXTRA ? 1 : 0);
            const int j0 = STEP * step0;                   // first memory position of this thread
            const int n0 = j0 - nold + 1;                  // recursion steps before it, counted from the frame's phase reference
            const int nold_run = nold - j0;                // positions of this run that are last frame's (<= 0: none, >= its length: all)
            const bool oldl = nold_run > 0;
            const int ninp = (int)s_prev[2 * M];
            const bool no_tail = ninp == 0;                // a stream's very first frame: integrator memory is zero
            // a step's 16 raw samples, two per dword: new ones from global memory -- 32 contiguous bytes per thread, fetched as two 16-byte
            // loads (sample by sample it was 16 loads per step whose 64 lanes each touched a different cache line: 4096 line accesses per
            // wave and frame, as much time in the texture path as the whole frame's arithmetic) -- old ones from the tail kept in LDS
            auto load_step = [&](int blk, uint32_t *dst) {
                const int jb = j0 + STEP * blk;
                if (jb >= nold) {
                    const PIRIP_GLOBAL char *p = (const PIRIP_GLOBAL char *)gin + 2u * (unsigned)(jb - nold);
#pragma unroll
                    for (int q = 0; q < STEP / 8; q++) {
                        const u32x4 v = *(const PIRIP_GLOBAL u32x4_a2 *)(p + 16 * q);
                        dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < STEP; k += 2) {
                        uint32_t pair = 0;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int j = jb + k + h;
                            const uint32_t w = j < nold ? (uint32_t)s_tail[HIST - nold + j] : (uint32_t)ldg(gin, (unsigned)(j < nold ? 0 : j - nold));
                            pair |= w << (16 * h);
                        }
                        dst[k / 2] = pair;
                    }
                }
            };
            // Tones two at a time: four tones' oscillators, switch values and sums at once are 230 VGPR (two workgroups per CU); a
            // second pass over the run converts its 64 samples again (+ 13 % instructions in this phase) and fits 168 (three).
            constexpr int TPP = 2;
#pragma unroll
            for (int mp = 0; mp < M; mp += TPP) {
                v2f ph[TPP], dph[TPP], swp[TPP], swd[TPP];
#pragma unroll
                for (int m = 0; m < TPP; m++) {
                    const uint32_t dthp = s_prev[mp + m], tixp = s_prev[M + mp + m];
                    const v2f stn = ldg(g_step, (unsigned)tix[mp + m]), stp = ldg(g_step, tixp);
                    const float dn = ldg(g_drift, (unsigned)tix[mp + m]).x, dp = ldg(g_drift, tixp).x;
                    const uint32_t th = (uint32_t)n0 * (oldl ? dthp : dthv[mp + m]);
                    const float g = 1.0f + (oldl ? dp * (float)(ninp + n0) : dn * (float)n0);
                    const v2f pcs = phasor(th);
                    ph[m] = v2f{pcs.x * g, pcs.y * g};
                    if (no_tail && oldl) ph[m] = v2f{0.f, 0.f};    // no integrator memory yet: old positions contribute zeros (a zero phasor stays zero)
                    dph[m] = oldl ? v2f{stp.x, stp.y} : v2f{stn.x, stn.y};
                    const v2f p1 = phasor(dthv[mp + m]);
                    const float g1 = 1.0f + dn;
                    swp[m] = v2f{p1.x * g1, p1.y * g1};
                    swd[m] = v2f{stn.x, stn.y};
                }
#pragma unroll 1
                for (int blk = 0; blk < CSTEPS + 1; blk++) {
                    if (blk >= nsteps) continue;
                    uint32_t rawv[STEP / 2];
                    load_step(blk, rawv);
                    v2f acc[TPP];
#pragma unroll
                    for (int m = 0; m < TPP; m++) acc[m] = v2f{0.f, 0.f};
                    // the first new sample of the frame (nold is a multiple of 4) is where the new oscillator starts: one thread of the
                    // workgroup meets it, in one of its steps -- every other wave-pass runs the loop without the test and its selects
                    const int sw_k = nold_run - STEP * blk;        // its place in this step
                    auto pass = [&](auto with_switch) {
#pragma unroll
                        for (int k = 0; k < STEP; k++) {
                            if constexpr (decltype(with_switch)::value) {
                                if ((k & 3) == 0 && k == sw_k) {
#pragma unroll
                                    for (int m = 0; m < TPP; m++) { ph[m] = swp[m]; dph[m] = swd[m]; }
                                }
                            }
                            const v2f x = (k & 1) ? cvt_sample_hi<FMT>(rawv[k >> 1]) : cvt_sample<FMT>(rawv[k >> 1]);
#pragma unroll
                            for (int m = 0; m < TPP; m++) {
                                acc[m] = mix_conj_acc(x, ph[m], acc[m]);
                                ph[m] = rot_step(ph[m], dph[m]);
                            }
                        }
                    };
                    if (__builtin_amdgcn_ballot_w64(oldl && sw_k >= 0 && sw_k < STEP)) pass(std::true_type{});
                    else pass(std::false_type{});
#pragma unroll
                    for (int m = 0; m < TPP; m++) s_step[mp + m][step0 + blk] = make_float2(acc[m].x, acc[m].y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#if PIRIP_BLOCK_PREFETCH
        // (issued right after the FFTs instead, a whole correlator pass earlier, it measured the same)
        // touch the next frame's cache lines (one dword of each 128-byte line, landing in a dump row of LDS nobody reads): they are on their
        // way to L2 while the window sums, the timing estimate and the decisions run, instead of being fetched from HBM by the next FFT
        {
            const int64_t left = nsamp - (pos + nin);                      // samples behind this frame
            const unsigned off = (unsigned)tid * 128u;
            if ((int64_t)(off / 2 + 2) <= left && off < (unsigned)(N + Q) * 2u)
                __builtin_amdgcn_global_load_lds((const PIRIP_GLOBAL void *)((const PIRIP_GLOBAL char *)gin + 2u * (unsigned)nin + off),
                                                 (__attribute__((address_space(3))) void *)s_dump, 4, 0, 0);
        }
#endif
        // the frame's last HIST raw samples are the next frame's old positions
        for (int i = tid; i < HIST; i += NT) s_tail[i] = ldg(gin, (unsigned)(nin - HIST + i));
        if (tid == 0) {                                    // (the correlator's reads are behind the barrier above; the next ones are a frame away)
            s_prev[2 * M] = (uint32_t)nin;
#pragma unroll
            for (int m = 0; m < M; m++) { s_prev[m] = dthv[m]; s_prev[M + m] = (uint32_t)tix[m]; }
        }
        // ================= a-7: window sums (15 steps each), fine timing ================================================
        float2 (*fint)[NINT] = (float2 (*)[NINT])(s_xa + M * NSTEP);
        float tcr = 0.f, tci = 0.f;
        {
            PIRIP_ARGS();
            const PIRIP_GLOBAL v2f *__restrict__ g_trec = gl((const v2f *)a.t.timing_rec);
            for (int w = tid; w < NINT; w += NT) {
                float ft1 = 0.f;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    v2f acc{0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < P; q++) { const float2 v = s_step[m][w + q]; acc = acc + v2f{v.x, v.y}; }
                    fint[m][w] = make_float2(acc.x, acc.y);
                    ft1 += (acc.x * acc.x) + (acc.y * acc.y);
                }
                const v2f tp = ldg(g_trec, (unsigned)w);
                tcr += ft1 * tp.x; tci += ft1 * tp.y;
            }
        }
        block_sum2(tcr, tci, s_red, tid);                  // (the barriers inside also publish fint)

        // ================= a-8 ========================================================================================
        {
            PIRIP_ARGS();
            const FskDims &d = a.d;
            const int frame_bytes = d.pack_bits ? (d.Nbits + 7) / 8 : d.Nbits;
            const size_t orow = (size_t)(frame + out0);
            PIRIP_GLOBAL uint8_t *bits_o = gl(a.io.bits ? a.io.bits + (size_t)sid * a.io.bits_stride + orow * frame_bytes : nullptr);
            PIRIP_GLOBAL float *filt_o = gl(a.io.filt ? a.io.filt + (size_t)sid * a.io.filt_stride + orow * M * NSYM : nullptr);
            PIRIP_GLOBAL float *stats_o = gl(a.io.stats ? a.io.stats + (size_t)sid * a.io.stats_stride + orow * PIRIP_STATS_PER_FRAME : nullptr);
            const bool bad = isnan(tcr) || isnan(tci);
            int nin_next = nin;
            if (!bad) {
                const float norm_rx_timing = atan2f(tci, tcr) * 0.15915494309189535f;
                const float rx_timing = norm_rx_timing * (float)P;
                const float d_norm = norm_rx_timing - sc_norm_rx_timing;
                sc_norm_rx_timing = norm_rx_timing;
                if (fabsf(d_norm) < 0.2f) {
                    const float appm = (1e6f * d_norm) / (float)NSYM;
                    sc_ppm = (0.9f * sc_ppm) + (0.1f * appm);
                }
                nin_next = N;
                if (!d.burst_mode) {
                    if (norm_rx_timing > 0.25f) nin_next = N + Q;
                    else if (norm_rx_timing < -0.25f) nin_next = N - Q;
                }
                const int low_sample = (int)floorf(rx_timing);
                const float fract = rx_timing - (float)low_sample;
                const int high_sample = (int)ceilf(rx_timing);
                float sig = 0.f, nse = 0.f, mean_e = 0.f, std_e = 0.f;
                const bool act = tid < NSYM;
                int sym = 0;
                float tmax[M];
                {
                    const int st = ((act ? tid : 0) + 1) * P;
                    float sum = 0.f;
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        const float2 lo = fint[m][st + low_sample], hi = fint[m][st + high_sample];
                        float2 t;
                        t.x = (1 - fract) * lo.x; t.y = (1 - fract) * lo.y;
                        t.x = t.x + fract * hi.x; t.y = t.y + fract * hi.y;
                        tmax[m] = (t.x * t.x) + (t.y * t.y);
                        sum += tmax[m];
                    }
                    float mx = tmax[0];
#pragma unroll
                    for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
                    if (act) { sig = mx; nse = (sum - mx) / (float)(M - 1); std_e = mx; mean_e = sqrtf(mx); }
                }
                if (bits_o && !d.pack_bits) {
                    if (act) {
                        if (M == 2) bits_o[tid] = sym == 1;
                        else { bits_o[2 * tid + 1] = sym & 1; bits_o[2 * tid] = (sym & 2) >> 1; }
                    }
                } else if (bits_o && tid < kWave) {        // wave 0 holds all Nsym decisions: ballots, lane j assembles byte j
                    const unsigned long long mlo = __ballot(act && (sym & 1)), mhi = __ballot(act && (sym & 2));
                    if (tid < frame_bytes) {
                        unsigned byte = 0;
                        if (M == 2) byte = __builtin_bitreverse32((unsigned)(mlo >> (8 * tid)) & 0xffu) >> 24;
                        else {
                            const unsigned h4 = (unsigned)(mhi >> (4 * tid)) & 0xfu, l4 = (unsigned)(mlo >> (4 * tid)) & 0xfu;
#pragma unroll
                            for (int q = 0; q < 4; q++) byte |= (((h4 >> q) & 1u) << (7 - 2 * q)) | (((l4 >> q) & 1u) << (6 - 2 * q));
                        }
                        bits_o[tid] = (uint8_t)byte;
                    }
                }
                if (act && filt_o) {
#pragma unroll
                    for (int m = 0; m < M; m++) filt_o[m * NSYM + tid] = sqrtf(tmax[m]);
                }
                // SNRest / EbNodB / snr_est / v_est / rx_*_pow: per-frame outputs and stream state that only the LAST frame of a call
                // leaves behind (the wave kernels' rule): computed on observable frames, by wave 0 alone (it holds all Nsym decisions,
                // thread 0 is the only one that writes them anywhere) -- no workgroup barrier
                const bool last_frame = (frame + 1 >= max_frames) || (pos + nin + nin_next > nsamp);
                if ((stats_o || last_frame) && tid < kWave) {
                    sig = wave_sum(sig); nse = wave_sum(nse) + 1e-12f;
                    mean_e = wave_sum(mean_e); std_e = wave_sum(std_e);
                    sig = sig / (float)NSYM; nse = nse / (float)NSYM;
                    const float v_est = sqrtf(sig - nse), SNRest = sig / nse;
                    mean_e = mean_e / (float)NSYM;
                    std_e = (std_e / (float)NSYM) - (mean_e * mean_e);
                    std_e = std_e > 0.0f ? sqrtf(std_e) : 0.0f;
                    const float EbNodB = -6.0f + (20.0f * log10f((1e-6f + mean_e) / (1e-6f + std_e)));
                    const float snr_est = (0.5f * s_sc[1]) + (0.5f * EbNodB);
                    if (tid == 0) { s_sc[0] = SNRest; s_sc[1] = snr_est; s_sc[2] = EbNodB; s_sc[3] = v_est; s_sc[4] = sig; s_sc[5] = nse; }
                }
            } else {
                // (a NaN frame; not unrolled or interleaved: by 8, the index vectors of these loops were hoisted out of the frame loop and spilled)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int i = tid; i < frame_bytes; i += NT) if (bits_o) bits_o[i] = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int i = tid; i < M * NSYM; i += NT) if (filt_o) filt_o[i] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < kMaxTones; m++) if (tid == 0) s_fest[m] = f_est[m];
            have_frames = true;
            if (stats_o && tid == 0) {
                stats_o[0] = f_est[0]; stats_o[1] = f_est[1]; stats_o[2] = f_est[2]; stats_o[3] = f_est[3];
                stats_o[4] = sc_norm_rx_timing; stats_o[5] = s_sc[0]; stats_o[6] = (float)nin_next; stats_o[7] = sc_ppm;
                stats_o[8] = bad ? 0.f : s_sc[4]; stats_o[9] = bad ? 0.f : s_sc[5];
            }
            pos += nin;
            nin = nin_next;
            frame++;
        }
        __syncthreads();
    }

    // ---- save stream state ---------------------------------------------------------------------------------------------
    {
        PIRIP_ARGS();
#pragma unroll
        for (int r = 0; r < 16; r++) gl(a.s.Sf + (size_t)sid * NDFT)[(tid + 256 * (r >> 2) + 1024 * (r & 3) + NDFT / 2) & (NDFT - 1)] = SfR[r];
        PIRIP_GLOBAL uint32_t *st32 = gl((uint32_t *)(a.s.hist + (size_t)sid * M * HIST));
        constexpr int TR = (HIST * 2 + 15) / 16 * 4;
        __syncthreads();
        for (int i = tid; i < HIST / 2; i += NT) st32[i] = ((const uint32_t *)s_tail)[i];
        if (tid == 0) {
#pragma unroll
            for (int m = 0; m < M; m++) { st32[TR + m] = s_prev[m]; st32[TR + M + m] = s_prev[M + m]; }
            st32[TR + 2 * M] = s_prev[2 * M];
            StreamScalars sc = a.s.scal[sid];
            sc.nin = nin; sc.norm_rx_timing = sc_norm_rx_timing; sc.ppm = sc_ppm; sc.SNRest = s_sc[0]; sc.snr_est = s_sc[1];
            sc.EbNodB = s_sc[2]; sc.v_est = s_sc[3]; sc.rx_sig_pow = s_sc[4]; sc.rx_nse_pow = s_sc[5];
            if (have_frames) for (int m = 0; m < kMaxTones; m++) sc.f_est[m] = s_fest[m];
            a.s.scal[sid] = sc;
            if (a.io.nframes) gl(a.io.nframes)[sid] = (int32_t)frame;
            if (a.io.consumed) gl(a.io.consumed)[sid] = pos;
        }
    }
#undef PIRIP_ARGS
}

// ---- instances and dispatch ----------------------------------------------------------------------------------------------
namespace {
struct BlockInst { int M, fmt, mask; void (*kern)(DemodArgs); };
#define PIRIP_BLOCK_INST(M, FMT, MASK) {M, FMT, MASK, fsk_demod_block_kernel<M, FMT, MASK>}
const BlockInst kBlockInst[] = {
    PIRIP_BLOCK_INST(2, PIRIP_IN_CU8_CSDR, false), PIRIP_BLOCK_INST(2, PIRIP_IN_CU8_CSDR, true),           // rtl_fsk -r 1000 (README.md:152,184)
    PIRIP_BLOCK_INST(4, PIRIP_IN_CU8_CSDR, false), PIRIP_BLOCK_INST(4, PIRIP_IN_CU8_CSDR, true),           // ... -m 4 --mask 2000 (README.md:239)
    PIRIP_BLOCK_INST(2, PIRIP_IN_CU8_FSKDEMOD, false), PIRIP_BLOCK_INST(2, PIRIP_IN_CU8_FSKDEMOD, true),   // fsk_demod -d -p 15 on the same signal
    PIRIP_BLOCK_INST(4, PIRIP_IN_CU8_FSKDEMOD, false), PIRIP_BLOCK_INST(4, PIRIP_IN_CU8_FSKDEMOD, true),
};
#undef PIRIP_BLOCK_INST
const BlockInst *find_block(const FskDims &d)
{
    if (!d.recalled_fast_ok) return nullptr;            // (the instances carry the recalled constants' default values: pirip_fsk_recalled)
    if (d.Ts != TS || d.P != P || d.Nsym != NSYM || d.Ndft != NDFT || d.fft_fma || d.est_band) return nullptr;
    for (const BlockInst &b : kBlockInst)
        if (b.M == d.M && b.fmt == d.in_format && b.mask == (d.freq_est_type != 0)) return &b;
    return nullptr;
}
}  // namespace

bool demod_block_applicable(const FskDims &d) { return find_block(d) != nullptr; }

int demod_block_describe(const FskDims &d, char *buf, size_t n)
{
    const BlockInst *b = find_block(d);
    if (!b) return 0;
    return snprintf(buf, n, "fsk_demod_block_kernel<M=%d,Ts=%d,P=%d,Nsym=%d,Ndft=%d,%s,%s256 threads/stream>", b->M, TS, P, NSYM, NDFT,
                    b->fmt == PIRIP_IN_CU8_CSDR ? "u8 csdr" : "u8 -d", b->mask ? "mask estimator," : "");
}

hipError_t launch_demod_block(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const BlockInst *b = find_block(a.d);
    if (!b) return hipErrorNotSupported;
    hipLaunchKernelGGL(b->kern, dim3(nstreams), dim3(NT), 0, stream, a);
    return hipGetLastError();
}

}  // namespace pirip
