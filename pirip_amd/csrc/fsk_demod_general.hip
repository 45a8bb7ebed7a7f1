// pirip_amd/csrc/fsk_demod_general.hip -- general-configuration FSK demodulator kernel (gfx950).
//
// One workgroup (64-256 lanes) owns one IQ stream and walks its frames in order, because codec2's
// demodulator is frame-serial: nin, the smoothed spectrum Sf, the tone estimates, the local
// oscillator phases and the integrator memory all chain from frame to frame
// [UPSTREAM-RECALLED codec2 fsk.c: fsk_demod_freq_est + fsk_demod_core; SURVEY.md 8a rows
//  a-1, a-4 ... a-8]. Parallelism comes from the batch of independent streams (one workgroup of two
// waves each, the read-only tables shared through L1/L2) and from the 128 lanes inside a frame.
// All per-frame intermediates (raw samples, FFT work array, grouped f_dc of the tone in hand,
// f_int) live in LDS; HBM sees the u8/s16 IQ stream once and the bits. Occupancy is LDS-bound at
// one wave per SIMD, so every loop issues its LDS reads in batches of four before the arithmetic
// and the next frame's input is read into registers a frame ahead. Per-frame cycle split measured
// with s_memtime (config 3, 1024-point FFTs): estimator 45 %, down-conversion + windows 38 %,
// staging 7 %, peak pick 4 %, timing 4 %, decisions 2 %.
//
// This kernel handles every configuration fsk_create_hbr() accepts (M in {2,4}, any Ts/P/Nsym,
// power-of-two Ndft, peak or mask estimator, four input formats). The specialised kernel in
// fsk_demod_wave.hip overtakes it for the reference's main command lines; this one stays as the
// on-device cross-check and as the path for every other configuration.
//
// Numerics contract (DESIGN.md "parity"): the frequency-estimator path (conversion, Hann,
// FFT butterflies, |X|, IIR, peak pick) performs the same float32 operations in the same
// order as the CPU restatement, so Sf and f_est are bit-identical; compiled with
// -ffp-contract=off so no multiply-add is fused behind our back. The down-conversion
// oscillator is NOT the upstream recursion (a 1200-step serial chain per tone): phases come
// from a 32-bit phase accumulator and the FFT twiddle table, so f_dc / f_int / rx_filt agree
// with the oracle to rounding (stated tolerance 1e-4 of the frame's peak magnitude), and the
// hard bits agree exactly wherever the decision margin exceeds that.
#include <hip/hip_runtime.h>
#include <cmath>

#include "../../include/pirip_hip.h"
#include "fsk_device.hpp"

namespace pirip {

namespace {

constexpr int kWave = 64;
#ifndef PIRIP_GEN_U
#define PIRIP_GEN_U 2
#endif
constexpr int kU = PIRIP_GEN_U;   // independent items per thread per pass in the batched loops (reads first, then arithmetic)

// Wave reductions on the VALU's DPP paths (row shifts, then row broadcasts; lane 63 ends up with the result, one v_readlane
// hands it to every lane) -- no LDS round trips (__shfl_xor is ds_bpermute: six of them and their waits per reduction).
#define PIRIP_GEN_DPP(op) \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        "s_nop 1\n\tv_readlane_b32 %1, %0, 63"
__device__ __forceinline__ float wave_sum(float v)
{
    // lanes whose DPP source is outside the row / masked keep their own partial sum: the row_shr steps build inclusive
    // prefix sums inside each row of 16, the broadcasts add the preceding rows' totals -- lane 63 holds the wave total
    int tot;
    asm(PIRIP_GEN_DPP("v_add_f32_dpp") : "+v"(v), "=s"(tot));
    return __builtin_bit_cast(float, tot);
}

// Ordering point for LDS traffic inside ONE wave (the streams of a workgroup never exchange data, and they
// run different numbers of frames, so a workgroup barrier inside the frame loop would be wrong): LDS
// instructions of a wave execute in order, only the compiler has to be kept from moving accesses across.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// arg-max with codec2's tie rule (first maximum wins, only values > 0 count)
// (v >= 0 and never NaN here: a candidate only ever replaces "best" by being larger than it, and best starts at 0 -- so the
//  maximum is six v_max_f32 with a DPP source, the winner the smallest index among the lanes that hold it)
__device__ __forceinline__ void wave_argmax(float &v, int &idx)
{
    float red = v;
    int smax, smin;
    asm(PIRIP_GEN_DPP("v_max_f32_dpp") : "+v"(red), "=s"(smax));
    int cand = (__builtin_bit_cast(int, v) == smax) ? idx : 0x7fffffff;
    asm(PIRIP_GEN_DPP("v_min_i32_dpp") : "+v"(cand), "=s"(smin));
    v = __builtin_bit_cast(float, smax);
    idx = smin;
}

// LDS, per stream (= per workgroup; the read-only tables -- twiddles, Hann, digit-reversal -- are read from
// global memory and live in L1/L2). Kept small so several streams fit a CU: the input
// stays in its raw form for the u8 / s16 formats (converted where it is used -- all three conversions are
// exact in one to three FMAs), the peak-picking work copy of Sf aliases the FFT work array, and the
// down-converted samples are held for ONE tone at a time, summed in groups (tones are processed in sequence;
// only their last hist_len samples persist).
struct Lds {
    float2 *in;      // [nin_max]  (f32 input)              }
    short2 *in16;    // [nin_max]  (s16 input, raw)         } one of the three
    uchar2 *in8;     // [nin_max]  (u8 input, raw)          }
    float2 *X;       // [Ndft]     FFT work array; Sfw (float [Ndft]) and the step sums alias it
    float *Sf;       // [Ndft]
    float2 *fdc;     // [Nmem/grp] down-converted samples of the tone being processed, summed in groups of grp
    float2 *hist;    // [M][hist_len/grp] last entries of every tone's fdc (carried between frames)
    float2 *fint;    // [M][nint]
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Long frames of 8-bit samples (Ts = 240: 24 KB per frame) are not staged: the estimator and every tone's down-conversion read them
// from global memory (L2: the frame is read 2 + M times) -- with the 24 KB and the aliasing below two streams fit a CU instead of one.
__host__ __device__ inline bool direct_input(const FskDims &d)
{
    const bool u8 = d.in_format == PIRIP_IN_CU8_FSKDEMOD || d.in_format == PIRIP_IN_CU8_CSDR;
    return u8 && (size_t)(d.N + d.nin_step) * sizeof(uchar2) > 16384;
}

__host__ __device__ inline size_t carve(const FskDims &d, Lds *l, char *base)
{
    size_t off = 0;
    const int nin_max = d.N + d.nin_step;
    const bool u8 = d.in_format == PIRIP_IN_CU8_FSKDEMOD || d.in_format == PIRIP_IN_CU8_CSDR;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    size_t o_in = take(direct_input(d) ? 16 : (u8 ? sizeof(uchar2) : d.in_format == PIRIP_IN_CS16 ? sizeof(short2) : sizeof(float2)) * nin_max);
    // X also stages packed bits (Nbits bytes)
    size_t xbytes = sizeof(float2) * d.Ndft;
    if (xbytes < (size_t)d.Nbits) xbytes = d.Nbits;
    size_t o_X = take(xbytes);
    size_t o_Sf = take(sizeof(float) * d.Ndft);
    // the down-converted samples of the tone in hand live in the FFT work array when they fit (the estimator is done with it by then,
    // the bit staging comes after the last tone)
    const size_t fdc_bytes = sizeof(float2) * (d.Nmem / d.grp);
    size_t o_fdc = fdc_bytes <= xbytes ? o_X : take(fdc_bytes);
    size_t o_hist = take(sizeof(float2) * d.M * (d.hist_len / d.grp));
    size_t o_fint = take(sizeof(float2) * d.M * d.nint);
    if (l) {
        l->in = (float2 *)(base + o_in); l->in16 = (short2 *)(base + o_in); l->in8 = (uchar2 *)(base + o_in); l->X = (float2 *)(base + o_X);
        l->Sf = (float *)(base + o_Sf); l->fdc = (float2 *)(base + o_fdc);
        l->hist = (float2 *)(base + o_hist); l->fint = (float2 *)(base + o_fint);
    }
    return off;
}

// exp(+j theta), theta in 2^-32 turns: top log2(Ndft) bits index the twiddle table
// (tw[k] = exp(-j 2 pi k/Ndft)), the remainder is a small-angle rotation.
__device__ __forceinline__ float2 phasor(uint32_t theta, const float2 *tw, int log2n)
{
    const uint32_t idx = theta >> (32 - log2n);
    const uint32_t low = theta & ((1u << (32 - log2n)) - 1u);
    float2 w = tw[idx];
    float c = w.x, s = -w.y;
    if (low) {
        const float b = (float)low * 1.4629180792671596e-9f;   // 2*pi / 2^32
        const float b2 = b * b;
        const float cb = 1.0f - b2 * (0.5f - b2 * (1.0f / 24.0f));
        const float sb = b * (1.0f - b2 * ((1.0f / 6.0f) - b2 * (1.0f / 120.0f)));
        const float c2 = c * cb - s * sb;
        const float s2 = s * cb + c * sb;
        c = c2; s = s2;
    }
    return make_float2(c, s);
}

// One FFT stage over the LDS work array, U butterflies per lane per pass: all reads of the pass are issued
// before the arithmetic and all writes after it, so a lone wave on its SIMD overlaps the LDS round trips
// (butterflies of a stage touch disjoint slots). Arithmetic and its order are kiss_fft's kf_bfly4 / kf_bfly2.
template <int U>
__device__ __forceinline__ void fft_stage_r4(float2 *X, const float2 *tw, int m, int fs, int nb, int tid, int NT)
{
    const int sh = 31 - __clz(m);
    for (int b0 = tid; b0 < nb; b0 += U * NT) {
        float2 f0[U], f1[U], f2[U], f3[U], t1[U], t2[U], t3[U];
        float2 *F[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int b = (b0 + u * NT < nb) ? b0 + u * NT : b0;
            const int g = b >> sh, k = b - (g << sh);
            F[u] = X + ((g * 4) << sh) + k;
            t1[u] = tw[k * fs]; t2[u] = tw[2 * k * fs]; t3[u] = tw[3 * k * fs];
            f0[u] = F[u][0]; f1[u] = F[u][m]; f2[u] = F[u][2 * m]; f3[u] = F[u][3 * m];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float2 s0, s1, s2, s3, s4, s5;
            s0.x = f1[u].x * t1[u].x - f1[u].y * t1[u].y; s0.y = f1[u].x * t1[u].y + f1[u].y * t1[u].x;
            s1.x = f2[u].x * t2[u].x - f2[u].y * t2[u].y; s1.y = f2[u].x * t2[u].y + f2[u].y * t2[u].x;
            s2.x = f3[u].x * t3[u].x - f3[u].y * t3[u].y; s2.y = f3[u].x * t3[u].y + f3[u].y * t3[u].x;
            s5.x = f0[u].x - s1.x; s5.y = f0[u].y - s1.y;
            f0[u].x += s1.x; f0[u].y += s1.y;
            s3.x = s0.x + s2.x; s3.y = s0.y + s2.y;
            s4.x = s0.x - s2.x; s4.y = s0.y - s2.y;
            f2[u].x = f0[u].x - s3.x; f2[u].y = f0[u].y - s3.y;
            f0[u].x += s3.x; f0[u].y += s3.y;
            f1[u].x = s5.x + s4.y; f1[u].y = s5.y - s4.x;
            f3[u].x = s5.x - s4.y; f3[u].y = s5.y + s4.x;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (u == 0 || b0 + u * NT < nb) { F[u][0] = f0[u]; F[u][m] = f1[u]; F[u][2 * m] = f2[u]; F[u][3 * m] = f3[u]; }
    }
}

template <int U>
__device__ __forceinline__ void fft_stage_r2(float2 *X, const float2 *tw, int m, int fs, int nb, int tid, int NT)
{
    const int sh = 31 - __clz(m);
    for (int b0 = tid; b0 < nb; b0 += U * NT) {
        float2 f0[U], f1[U], t1[U];
        float2 *F[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int b = (b0 + u * NT < nb) ? b0 + u * NT : b0;
            const int g = b >> sh, k = b - (g << sh);
            F[u] = X + ((g * 2) << sh) + k;
            t1[u] = tw[k * fs];
            f0[u] = F[u][0]; f1[u] = F[u][m];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float2 t;
            t.x = f1[u].x * t1[u].x - f1[u].y * t1[u].y; t.y = f1[u].x * t1[u].y + f1[u].y * t1[u].x;
            f1[u].x = f0[u].x - t.x; f1[u].y = f0[u].y - t.y;
            f0[u].x += t.x; f0[u].y += t.y;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (u == 0 || b0 + u * NT < nb) { F[u][0] = f0[u]; F[u][m] = f1[u]; }
    }
}

// Workgroup reductions (one stream = one workgroup of NT/64 waves): wave step on the VALU, then <= kMaxWaves partials
// through LDS. Every thread gets the same result. red points at 2 * kMaxWaves words of LDS scratch.
constexpr int kMaxWaves = 16;
constexpr int kRedBytes = 2 * kMaxWaves * 4;
__device__ __forceinline__ float block_sum(float v, float *red, int tid, int NT)
{
    v = wave_sum(v);
    if (NT == kWave) return v;
    __syncthreads();
    if ((tid & (kWave - 1)) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (NT >> 6); w++) r += red[w];
    return r;
}
__device__ __forceinline__ void block_argmax(float &v, int &idx, float *red, int tid, int NT)
{
    wave_argmax(v, idx);
    if (NT == kWave) return;
    __syncthreads();
    if ((tid & (kWave - 1)) == 0) { red[tid >> 6] = v; ((int *)red)[kMaxWaves + (tid >> 6)] = idx; }
    __syncthreads();
    v = red[0]; idx = ((int *)red)[kMaxWaves];
    for (int w = 1; w < (NT >> 6); w++) {
        const float ov = red[w]; const int oi = ((int *)red)[kMaxWaves + w];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

}  // namespace

// Two builds of one body: MAXW = 4 (64..256 threads per stream, register budget of three waves per SIMD: measured best of
// {2, 3, 4} x {128, 256}) for frames of up to a few thousand samples, MAXW = 8 (up to 512 threads, two waves per SIMD, 256 VGPRs each) for long frames whose LDS
// footprint leaves one stream per CU anyway (Ts = 240 / Ndft = 4096: 111-126 KB) -- there the only occupancy is waves per stream.
// atan2f as glibc computes it (fdlibm e_atan2f.c / s_atanf.c: float operations only -- a division, two polynomial halves in Horner
// form, table offsets), restated for the exact first frame: the oracle's fine-timing angle is libm's atan2f, and ocml's differs from it
// in the last bit for some arguments. tests/test_gpu_parity.py compares this restatement with the host's atan2f on 10^7 arguments.
__device__ inline float glibc_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = __builtin_bit_cast(int32_t, x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = ((2.0f * x) - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + (1.5f * x)); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + (w * (aT[2] + (w * (aT[4] + (w * (aT[6] + (w * (aT[8] + (w * aT[10]))))))))));
    const float s2 = w * (aT[1] + (w * (aT[3] + (w * (aT[5] + (w * (aT[7] + (w * aT[9]))))))));
    if (id < 0) return x - (x * (s1 + s2));
    const float r = atanhi[id] - (((x * (s1 + s2)) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
__device__ inline float glibc_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = __builtin_bit_cast(int32_t, x), hy = __builtin_bit_cast(int32_t, y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return glibc_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? (3.0f * pi_o_4) + tiny : (-3.0f * pi_o_4) - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + (0.5f * pi_lo);
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = glibc_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// EXACT0: the first frame of a stream after fsk_create / reset with every operation in the ORACLE's order (fsk_oracle.c:300-400): the
// oscillator as the serial recursion phi_c *= dphi from (1, 0), the windows summed forward, the timing phasor sum accumulated serially
// in window order, atan2f as glibc's, the division by 2 pi in double. A recording that starts one sample before a symbol boundary
// hands the first decision (P == Ts) ONE sample; both tone magnitudes are then equal up to float rounding and the decision follows
// the last bit of the timing estimate -- only the same operations in the same order reproduce it. One frame, one
// launch, once per stream: cost does not matter; the demodulator proper (wave / block / this kernel) takes over at io.first.
template <int MAXW, int EXACT>
__device__ __forceinline__ void fsk_demod_general_body(const DemodArgs &a)
{
    // EXACT: 0 = the kernel proper; 1 = the exact first frame (prologue, "EXACT0" below); 2 = EVERY frame in the oracle's operation order
    // (PIRIP_KERNEL=exact: the on-device proof that the fast kernels' differences under noise are evaluation order and nothing else --
    // bits, soft magnitudes, timing, nin, SNRest bit for bit at any SNR; one thread walks the oscillator recursion and the timing sum of
    // every frame, ~0.15 ms per frame and stream: a checker's speed, not a receiver's)
    constexpr bool EXACT0 = EXACT == 1, EXACTM = EXACT != 0, EXACTALL = EXACT == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FskDims &d = a.d;
    const int M = d.M, Ndft = d.Ndft, Nmem = d.Nmem, nint = d.nint, Ts = d.Ts, P = d.P, Nsym = d.Nsym;
    const int log2n = 31 - __clz(Ndft);

    // One workgroup (NT = 64..256 threads) per stream: all its waves run the same frame loop and meet at
    // workgroup barriers. The read-only tables stay in global memory (L1/L2-resident, shared by every stream);
    // with 3-4 waves per SIMD their latency is covered, and LDS holds only per-stream data.
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const int sid = blockIdx.x;
    Lds L;
    float *red = (float *)smem;                            // 2 * kMaxWaves words of reduction scratch
    carve(d, &L, smem + kRedBytes);
    const float2 *__restrict__ g_tw = a.t.tw;
    const float *__restrict__ g_hann = a.t.hann;
    const uint16_t *__restrict__ g_perm = a.t.perm;

    // ---- stream state into LDS --------------------------------------------------------------
    const bool in_u8 = d.in_format == PIRIP_IN_CU8_FSKDEMOD || d.in_format == PIRIP_IN_CU8_CSDR;
    const bool in_s16 = d.in_format == PIRIP_IN_CS16;
    // exact u8 -> float: fsk_demod -d is (x - 127)/128 = fma(x, 2^-7, -127/128); csdr convert_u8_f is
    // x/127.5 - 1 (double, rounded) = fma(x, c_lo, fma(x, c_hi, -1)) with c_hi a multiple of 2^-22 (see decim_kernels.hip)
    const float cv_hi = d.in_format == PIRIP_IN_CU8_FSKDEMOD ? 0.0078125f : 0.007843255996704102f;
    const float cv_lo = d.in_format == PIRIP_IN_CU8_FSKDEMOD ? 0.0f : -1.187418e-07f;
    const float cv_c = d.in_format == PIRIP_IN_CU8_FSKDEMOD ? -0.9921875f : -1.0f;
    const bool direct = direct_input(d);
    const uchar2 *gin8 = nullptr;                          // direct input: this frame's samples in global memory
    auto sample = [&](int i) -> float2 {
        if (in_u8) {
            const uchar2 v = direct ? gin8[i] : L.in8[i];
            if (d.u8_table) return make_float2(a.t.lut[v.x], a.t.lut[v.y]);            // a -d map other than the recalled (x - 127) / 128: the plan's table
            const float xr = (float)v.x, xi = (float)v.y;
            return make_float2(__builtin_fmaf(xr, cv_lo, __builtin_fmaf(xr, cv_hi, cv_c)),
                               __builtin_fmaf(xi, cv_lo, __builtin_fmaf(xi, cv_hi, cv_c)));
        }
        if (in_s16) {
            // x / FDMDV_SCALE, correctly rounded, without the divide: q = x*r, then one Newton step on the
            // residual; equal to the IEEE quotient for every int16 value (checked exhaustively, tests/)
            const short2 v = L.in16[i];
            const float xr = (float)v.x, xi = (float)v.y;
            // (the divisor is plan data: 750 as recalled; the exhaustive check runs for 750 and for 1000)
            const float r = 1.0f / d.s16_scale;
            float qr = xr * r, qi = xi * r;
            qr = __builtin_fmaf(__builtin_fmaf(-d.s16_scale, qr, xr), r, qr);
            qi = __builtin_fmaf(__builtin_fmaf(-d.s16_scale, qi, xi), r, qi);
            return make_float2(qr, qi);
        }
        return L.in[i];
    };
    float *Sfw = (float *)L.X;
    for (int i = tid; i < Ndft; i += NT) L.Sf[i] = a.s.Sf[(size_t)sid * Ndft + i];
    const int G = d.grp, hist_g = d.hist_len / G, ng = Nmem / G;
    for (int m = 0; m < M; m++)
        for (int h = tid; h < hist_g; h += NT)
            L.hist[m * hist_g + h] = EXACT0 ? make_float2(0.f, 0.f) : a.s.hist[((size_t)sid * M + m) * d.hist_len + h];   // (a created stream's integrator
                                                                                     // memory is zero, whatever layout the handle's kernel keeps in this block)

    StreamScalars sc = a.s.scal[sid];
    uint32_t theta[kMaxTones];
#pragma unroll
    for (int m = 0; m < kMaxTones; m++) theta[m] = EXACT0 ? 0u : a.s.theta[(size_t)sid * kMaxTones + m];
    __syncthreads();

    const uint8_t *in_base = a.io.in + (size_t)sid * a.io.in_stride;
    constexpr int kPre = 12;                              // input read-ahead registers per thread (12 x NT samples)
    const int nin_max = d.N + d.nin_step;
    const bool can_pre = !EXACTM && !direct && (in_u8 || in_s16) && nin_max <= kPre * NT;
    bool have_pre = false;
    uint32_t pre[kPre];
    int64_t pos = 0;
    int64_t frame = 0;
    int nin = sc.nin;
    float2 phc[kMaxTones];                                // EXACT == 2: phi_c of every tone, carried frame to frame by thread 0
#pragma unroll
    for (int m = 0; m < kMaxTones; m++) phc[m] = (EXACTALL && a.s.phic) ? a.s.phic[(size_t)sid * kMaxTones + m] : make_float2(0.f, 0.f);
    uint32_t x0_dth[kMaxTones] = {0u, 0u, 0u, 0u};       // EXACT0: the frame's tone estimates as the wave kernel's state block names them
    int x0_tix[kMaxTones] = {0, 0, 0, 0};
    if (!EXACT0 && a.io.first) { pos = a.io.first[sid]; frame = pos ? 1 : 0; }      // an exact-first-frame prologue ran in this call
    const int64_t frame_limit = EXACT0 ? (a.io.max_frames < 1 ? a.io.max_frames : 1) : a.io.max_frames;

    while (frame < frame_limit && pos + nin <= a.io.nsamp) {
        // ---- a-1: convert nin samples to complex float -----------------------------------
        // The u8 / s16 input of the NEXT frame is requested into registers right after this frame's input has
        // landed in LDS (nin_max samples from pos + nin: the next nin is not known yet) and written to LDS at the
        // top of the next iteration, so the HBM round trip is covered by a whole frame of work. f32 input, or
        // frames longer than kPre*64 samples, are staged synchronously (eight loads per lane in flight).
        auto stage = [&](auto *dst, const auto *src) {
            for (int i0 = tid; i0 < nin; i0 += 8 * NT) {
                decltype(src[0] + src[0]) v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = i0 + u * NT; v[u] = src[i < nin ? i : nin - 1]; }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = i0 + u * NT; if (i < nin) dst[i] = v[u]; }
            }
        };
        if (direct) gin8 = (const uchar2 *)(in_base + 2 * pos);
        else if (have_pre) {
#pragma unroll
            for (int u = 0; u < kPre; u++) {
                const int i = tid + u * NT;
                if (i < nin) { if (in_u8) ((uint16_t *)L.in8)[i] = (uint16_t)pre[u]; else ((uint32_t *)L.in16)[i] = pre[u]; }
            }
        } else if (in_u8) stage((uint16_t *)L.in8, (const uint16_t *)(in_base + 2 * pos));
        else if (in_s16) stage((uint32_t *)L.in16, (const uint32_t *)((const short2 *)in_base + pos));
        else stage((double *)L.in, (const double *)((const float2 *)in_base + pos));
        __syncthreads();
        if (can_pre) {
            const int64_t p1 = pos + nin;
#pragma unroll
            for (int u = 0; u < kPre; u++) {
                const int i = tid + u * NT;
                if (u * NT < nin_max) {                     // wave-uniform
                    const int64_t gi = (p1 + i < a.io.nsamp) ? p1 + i : a.io.nsamp - 1;
                    pre[u] = in_u8 ? (uint32_t)((const uint16_t *)in_base)[gi] : ((const uint32_t *)in_base)[gi];
                }
            }
            have_pre = true;
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- a-5: frequency estimator ------------------------------------------------------
        const int numffts = nin / (Ndft / 2) - 1;
        for (int j = 0; j < numffts; j++) {
            const int off = j * Ndft / 2;
            // (loops below: kU independent items per thread per pass, reads first -- see fft_stage_r4)
            for (int i0 = tid; i0 < Ndft; i0 += kU * NT) {   // natural order in, digit-reversed slot out
                float h[kU]; float2 x[kU]; int dst[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int i = (i0 + u * NT < Ndft) ? i0 + u * NT : i0;
                    h[u] = g_hann[i]; x[u] = sample(off + i); dst[u] = g_perm[i];
                }
#pragma unroll
                for (int u = 0; u < kU; u++)
                    if (u == 0 || i0 + u * NT < Ndft) L.X[dst[u]] = make_float2(h[u] * x[u].x, h[u] * x[u].y);
            }
            __syncthreads();
            for (int s = 0; s < d.nstages; s++) {
                const int p = a.stages[s].radix, m = a.stages[s].m, fs = a.stages[s].fstride;
                const int nb = Ndft / p;
                if (p == 4) { if (nb >= kU * NT) fft_stage_r4<kU>(L.X, g_tw, m, fs, nb, tid, NT); else fft_stage_r4<1>(L.X, g_tw, m, fs, nb, tid, NT); }
                else { if (nb >= kU * NT) fft_stage_r2<kU>(L.X, g_tw, m, fs, nb, tid, NT); else fft_stage_r2<1>(L.X, g_tw, m, fs, nb, tid, NT); }
                __syncthreads();
            }
            // fftshift + |X| + first-order smoothing
            for (int i0 = tid; i0 < Ndft; i0 += kU * NT) {
                float2 x[kU]; float sf[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int i = (i0 + u * NT < Ndft) ? i0 + u * NT : i0;
                    x[u] = L.X[(i + Ndft / 2) & (Ndft - 1)]; sf[u] = L.Sf[i];
                }
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const float mag2 = (x[u].x * x[u].x) + (x[u].y * x[u].y);
                    if (u == 0 || i0 + u * NT < Ndft) L.Sf[i0 + u * NT] = (sf[u] * d.one_minus_tc) + ((d.sf_power ? mag2 : sqrtf(mag2)) * d.tc);
                }
            }
            __syncthreads();
        }

        // peak method (always run: f_est is reported even when the mask method drives the demod)
        int freqi[kMaxTones];
        for (int i = tid; i < Ndft; i += NT) Sfw[i] = L.Sf[i];
        __syncthreads();
        // (tone loops are written "unrolled over kMaxTones, guarded by m < M" wherever they index a per-tone register array:
        //  with a run-time trip count the arrays would live in scratch memory)
#pragma unroll
        for (int m = 0; m < kMaxTones; m++) if (m < M) {
            float best = 0.0f; int ib = 0;
            for (int j = d.est_st + tid; j < d.est_en; j += NT) {
                const float v = Sfw[j];
                if (v > best) { best = v; ib = j; }
            }
            block_argmax(best, ib, red, tid, NT);
            int f_min = ib - d.f_zero; f_min = f_min < 0 ? 0 : f_min;
            int f_max = ib + d.f_zero; f_max = f_max > Ndft ? Ndft : f_max;
            __syncthreads();
            for (int j = f_min + tid; j < f_max; j += NT) Sfw[j] = 0.0f;
            __syncthreads();
            freqi[m] = ib - Ndft / 2;
        }
        // ascending sort of M <= 4 indices (insertion sort, fully unrolled)
#pragma unroll
        for (int x = 1; x < kMaxTones; x++)
#pragma unroll
            for (int y = x; y > 0; y--)
                if (x < M && freqi[y] < freqi[y - 1]) { const int t = freqi[y]; freqi[y] = freqi[y - 1]; freqi[y - 1] = t; }
        float f_est[kMaxTones] = {0.f, 0.f, 0.f, 0.f};
        uint32_t dtheta[kMaxTones] = {0u, 0u, 0u, 0u};
        int drift_ix[kMaxTones] = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < kMaxTones; m++) if (m < M) {
            f_est[m] = (float)freqi[m] * d.bin_hz;
            dtheta[m] = (uint32_t)freqi[m] << (32 - log2n);
            drift_ix[m] = freqi[m] + Ndft / 2;
        }
        if (d.freq_est_type) {
            // mask method: comb of 3-bin teeth slid over Sf, tooth sums in ascending order
            float best = 0.0f; int bb = d.est_st;
            for (int b = d.est_st + tid; b < d.est_en - d.mask_len; b += NT) {
                float corr = 0.0f;
                for (int k = 0; k < d.n_teeth; k++) corr += L.Sf[b + a.t.teeth[k]];
                if (corr > best) { best = corr; bb = b; }
            }
            // lanes that found nothing keep (0, est_st): smallest index wins ties, as upstream
            block_argmax(best, bb, red, tid, NT);
            const float foff = (float)((bb - Ndft / 2) * d.Fs / Ndft);
#pragma unroll
            for (int m = 0; m < kMaxTones; m++) if (m < M) {
                f_est[m] = foff + (float)(m * d.tone_spacing);
                dtheta[m] = a.t.mask_dtheta[bb * M + m];
                drift_ix[m] = bb * M + m;
            }
        }

        if (EXACT0) { for (int m = 0; m < kMaxTones; m++) { x0_dth[m] = dtheta[m]; x0_tix[m] = drift_ix[m]; } }
        // ---- a-6: shift integrator memory, down-convert, integrate -------------------------
        const int nold = Nmem - nin;
        // One tone at a time: the last nold samples of the previous frame come back from hist, the new samples
        // are down-converted behind them, the tail is saved for the next frame, then the windows are summed.
        // Integrator memory is kept as sums over groups of G samples (G = Ts/P when the frame shifts are whole
        // groups, else 1): window i = P (or Ts) consecutive entries. A lane down-converts a run of consecutive
        // samples: one table phasor at the run start, then the upstream's own rounded per-sample multiplier.
        const int nold_g = nold / G, per_win = Ts / G, win_step = (Ts / P) / G;
        const int run = G > 1 ? G : (nin + NT - 1) / NT;   // samples per lane run (G: one stored entry per run)
#pragma unroll
        for (int m = 0; m < kMaxTones; m++) if (m < M) {
            for (int i = tid; i < nold_g; i += NT) L.fdc[i] = L.hist[m * hist_g + hist_g - nold_g + i];
            const uint32_t th0 = theta[m], dth = dtheta[m];
            // upstream advances phi_c by a float32-rounded multiplier, so |phi_c| drifts as
            // (1+a)^n inside a frame (renormalised at its end): track that gain to first order
            const float gain_slope = a.t.osc_drift[drift_ix[m]].x;
            const float2 rot = a.t.osc_step[drift_ix[m]];
            if (EXACTM) {
                // [fsk_oracle.c:316-320] phi_c[m] = cmult(phi_c[m], dphi_m); f_dc = cmult(in, cconj(phi_c[m])) -- one thread, sample by sample
                // (G == 1 here: demod_exact0_applicable). phi_c starts at (1, 0): the prologue only runs on a stream in its created state.
                if (tid == 0) {
                    float2 ph = EXACTALL ? phc[m] : make_float2(1.0f, 0.0f);
                    if (EXACTALL && ph.x == 0.0f && ph.y == 0.0f) ph = make_float2(1.0f, 0.0f);    // the created state (a normalised phasor is never 0)
                    for (int j = 0; j < nin; j++) {
                        const float2 np = make_float2((ph.x * rot.x) - (ph.y * rot.y), (ph.x * rot.y) + (ph.y * rot.x));
                        ph = np;
                        const float2 x = sample(j);
                        const float ci = -ph.y;                       // cconj
                        L.fdc[nold + j] = make_float2((x.x * ph.x) - (x.y * ci), (x.x * ci) + (x.y * ph.x));
                    }
                    if (EXACTALL) {                                   // phi_c[m] = comp_normalize(phi_c[m]), carried to the next frame
                        const float av = sqrtf((ph.x * ph.x) + (ph.y * ph.y));
                        phc[m] = make_float2(ph.x / av, ph.y / av);
                    }
                }
            } else
            for (int j0 = tid * run; j0 < nin; j0 += NT * run) {
                float2 ph = phasor(th0 + (uint32_t)(j0 + 1) * dth, g_tw, log2n);
                float2 acc = make_float2(0.f, 0.f);
                const int j1 = (j0 + run < nin) ? j0 + run : nin;
                const float g = 1.0f + gain_slope * (float)(j0 + 1);   // drift up to the run start; the recursion adds its own
                for (int jb = j0; jb < j1; jb += 4) {
                    float2 xs[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) xs[u] = sample(jb + u < j1 ? jb + u : j1 - 1);   // reads first
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = jb + u;
                        if (j < j1) {
                            const float2 x = xs[u];
                            const float2 y = make_float2((x.x * ph.x + x.y * ph.y) * g, (x.y * ph.x - x.x * ph.y) * g);
                            if (G > 1) { acc.x += y.x; acc.y += y.y; }
                            else L.fdc[nold + j] = y;
                            const float2 nph = make_float2(ph.x * rot.x - ph.y * rot.y, ph.x * rot.y + ph.y * rot.x);
                            ph = nph;
                        }
                    }
                }
                if (G > 1) L.fdc[nold_g + j0 / G] = acc;
            }
            theta[m] = th0 + (uint32_t)nin * dth;
            __syncthreads();
            for (int h = tid; h < hist_g; h += NT) L.hist[m * hist_g + h] = L.fdc[ng - hist_g + h];
            for (int i = tid; i < nint; i += NT) {
                const float2 *src = L.fdc + i * win_step;
                float2 acc = make_float2(0.f, 0.f);
                for (int q = 0; q < per_win; q++) { acc.x += src[q].x; acc.y += src[q].y; }
                L.fint[m * nint + i] = acc;
            }
            __syncthreads();
        }

        // ---- a-7: fine timing -----------------------------------------------------------------
        float tcr = 0.f, tci = 0.f;
        for (int i = tid; i < nint; i += NT) {
            float ft1 = 0.f;
            for (int m = 0; m < M; m++) {
                const float2 v = L.fint[m * nint + i];
                ft1 += (v.x * v.x) + (v.y * v.y);
            }
            const float2 ph = a.t.timing_rec[i];   // the upstream recursion's phasor, drift included
            tcr += ft1 * ph.x; tci += ft1 * ph.y;
        }
        if (EXACTM) {
            // [fsk_oracle.c:338-348] ft1 per window in parallel (same operations), the accumulation t_c += ft1 * phi_ft serially in window order
            float *ft1s = (float *)L.fdc;                   // the last tone's down-converted samples are dead; nint floats fit in Nmem float2
            __syncthreads();
            for (int i = tid; i < nint; i += NT) {
                float ft1 = 0.f;
                for (int m = 0; m < M; m++) {
                    const float2 v = L.fint[m * nint + i];
                    ft1 += (v.x * v.x) + (v.y * v.y);
                }
                ft1s[i] = ft1;
            }
            __syncthreads();
            if (tid == 0) {
                float tr = 0.f, ti = 0.f;
                for (int i = 0; i < nint; i++) {
                    const float2 ph = a.t.timing_rec[i];
                    tr = tr + (ft1s[i] * ph.x); ti = ti + (ft1s[i] * ph.y);
                }
                red[0] = tr; red[1] = ti;
            }
            __syncthreads();
            tcr = red[0]; tci = red[1];
            __syncthreads();
        } else { tcr = block_sum(tcr, red, tid, NT); tci = block_sum(tci, red, tid, NT); }

        const int frame_bytes = d.pack_bits ? (d.Nbits + 7) / 8 : d.Nbits;
        uint8_t *bits_o = a.io.bits ? a.io.bits + (size_t)sid * a.io.bits_stride + (size_t)frame * frame_bytes : nullptr;
        uint8_t *bits_l = (uint8_t *)L.X;            // FFT work array is free here: staging for packed output (Nbits <= 8*Ndft)
        float *filt_o = a.io.filt ? a.io.filt + (size_t)sid * a.io.filt_stride + (size_t)frame * M * Nsym : nullptr;
        float *stats_o = a.io.stats ? a.io.stats + (size_t)sid * a.io.stats_stride + (size_t)frame * PIRIP_STATS_PER_FRAME : nullptr;

        const bool bad = isnan(tcr) || isnan(tci);
        if (!bad) {
            // (single precision, as in the wave kernel: codec2 divides by 2 pi and smooths ppm in double; the results differ by at
            //  most an ulp, far inside what the summation order already moves them, and double-precision code in this
            //  once-per-frame block costs registers for the whole kernel)
            // (EXACT0: atan2f(t_c.imag, t_c.real) / (2 * M_PI) as C evaluates it: glibc's atan2f, the division in double)
            const float norm_rx_timing = EXACTM ? (float)((double)glibc_atan2f(tci, tcr) / (2 * 3.14159265358979323846)) : atan2f(tci, tcr) * 0.15915494309189535f;
            const float rx_timing = norm_rx_timing * (float)P;
            const float d_norm = norm_rx_timing - sc.norm_rx_timing;
            sc.norm_rx_timing = norm_rx_timing;
            if (fabsf(d_norm) < 0.2f) {
                if (EXACTALL) {                                       // [fsk_oracle.c] appm = 1e6 * d / (float)nsym; ppm = .9 * ppm + .1 * appm: double arithmetic, float stores
                    const float appm = (float)(1e6 * (double)d_norm / (double)(float)Nsym);
                    sc.ppm = (float)(.9 * (double)sc.ppm + .1 * (double)appm);
                } else {
                    const float appm = (1e6f * d_norm) / (float)Nsym;
                    sc.ppm = (0.9f * sc.ppm) + (0.1f * appm);
                }
            }
            int nin_next = d.N;
            if (!d.burst_mode) {
                if (norm_rx_timing > d.nin_thresh) nin_next = d.N + d.nin_step;         // (plan: 0.25 and Ts / 4 as recalled)
                else if (norm_rx_timing < -d.nin_thresh) nin_next = d.N - d.nin_step;
            }

            // ---- a-8: resample, decide, stats ------------------------------------------------
            const int low_sample = (int)floorf(rx_timing);
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = (int)ceilf(rx_timing);
            // ---- MODEM_STATS.rx_eye (diagnostic output of fsk_get_demod_stats / fsk_demod -t; off unless asked for) ------------
            // [UPSTREAM-RECALLED fsk.c, end of fsk_demod_core]: ET_MAX / M traces per tone, two symbols of |f_int| each, every dec-th
            // position kept, trace i of tone m in row i*M + m from position 2P(i + 1) + neyeoffset, neyeoffset = high_sample + 1
            // (the trace centred on the timing estimate); unnormalised here (the reader normalises).
            if (a.io.eye) {
                const int dec = (2 * P + kEyePoints - 1) / kEyePoints, npts = (2 * P) / dec;
                int traces = kEyeTraces / M;
                while (traces > 0 && 2 * P * traces + (P / 2 + 1) + (npts - 1) * dec >= nint) traces--;
                float *eo = a.io.eye + (size_t)sid * kEyeTraces * kEyePoints;
                for (int e = tid; e < traces * M * npts; e += NT) {
                    const int row = e / npts, j = e - row * npts, i = row / M, m = row - i * M;
                    const float2 v = L.fint[m * nint + 2 * P * (i + 1) + (high_sample + 1) + dec * j];
                    eo[row * kEyePoints + j] = sqrtf((v.x * v.x) + (v.y * v.y));
                }
            }
            float sig = 0.f, nse = 0.f, mean_e = 0.f, std_e = 0.f;
            for (int i = tid; i < Nsym; i += NT) {
                const int st = (i + 1) * P;
                float tmax[kMaxTones] = {0.f, 0.f, 0.f, 0.f};
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < kMaxTones; m++) if (m < M) {
                    const float2 lo = L.fint[m * nint + st + low_sample];
                    const float2 hi = L.fint[m * nint + st + high_sample];
                    float2 t;
                    t.x = (1 - fract) * lo.x; t.y = (1 - fract) * lo.y;
                    t.x = t.x + fract * hi.x; t.y = t.y + fract * hi.y;
                    tmax[m] = (t.x * t.x) + (t.y * t.y);
                    sum += tmax[m];
                }
                float mx = tmax[0]; int sym = 0;
#pragma unroll
                for (int m = 1; m < kMaxTones; m++) if (m < M && tmax[m] > mx) { mx = tmax[m]; sym = m; }
                if (bits_o) {
                    uint8_t *bo = d.pack_bits ? bits_l : bits_o;
                    if (M == 2) bo[i] = sym == 1;
                    else { bo[2 * i + 1] = sym & 1; bo[2 * i] = (sym & 2) >> 1; }
                }
                if (filt_o) {
#pragma unroll
                    for (int m = 0; m < kMaxTones; m++) if (m < M) filt_o[m * Nsym + i] = sqrtf(tmax[m]);
                }
                sig += mx;
                nse += (sum - mx) / (float)(M - 1);
                std_e += mx;
                mean_e += sqrtf(mx);
                if (EXACTALL) { ((float *)L.fdc)[i] = mx; ((float *)L.fdc)[Nsym + i] = (sum - mx) / (float)(M - 1); }   // (the timing terms there are dead)
            }
            if (bits_o && d.pack_bits) {
                __syncthreads();
                for (int j = tid; j < frame_bytes; j += NT) {
                    unsigned byte = 0;
                    for (int b = 0; b < 8; b++) if (8 * j + b < d.Nbits) byte |= (unsigned)(bits_l[8 * j + b] & 1) << (7 - b);
                    bits_o[j] = (uint8_t)byte;
                }
            }
            sig = block_sum(sig, red, tid, NT); nse = block_sum(nse, red, tid, NT) + 1e-12f;
            mean_e = block_sum(mean_e, red, tid, NT); std_e = block_sum(std_e, red, tid, NT);
            if (EXACTALL) {
                // [fsk_oracle.c] rx_sig_pow, rx_nse_pow (seeded 1e-12), meanebno, stdebno accumulated serially in symbol order
                __syncthreads();
                if (tid == 0) {
                    float rs = 0.0f, rn = 1e-12f, me = 0.0f, se = 0.0f;
                    for (int i = 0; i < Nsym; i++) {
                        const float mxv = ((float *)L.fdc)[i];
                        rs += mxv; rn += ((float *)L.fdc)[Nsym + i];
                        se += mxv; me += sqrtf(mxv);
                    }
                    red[0] = rs; red[1] = rn; red[2] = me; red[3] = se;
                }
                __syncthreads();
                sig = red[0]; nse = red[1]; mean_e = red[2]; std_e = red[3];
                __syncthreads();
            }
            sig = sig / (float)Nsym; nse = nse / (float)Nsym;
            sc.v_est = sqrtf(sig - nse);
            sc.SNRest = sig / nse;
            sc.rx_sig_pow = sig; sc.rx_nse_pow = nse;
            mean_e = mean_e / (float)Nsym;
            std_e = (std_e / (float)Nsym) - (mean_e * mean_e);
            std_e = std_e > 0.0f ? sqrtf(std_e) : 0.0f;
            sc.EbNodB = -6.0f + (20.0f * log10f((1e-6f + mean_e) / (1e-6f + std_e)));
            sc.snr_est = (0.5f * sc.snr_est) + (0.5f * sc.EbNodB);
            nin = nin_next;
        } else {
            // NaN in the timing estimate: upstream returns before touching the outputs
            for (int i = tid; i < frame_bytes; i += NT) if (bits_o) bits_o[i] = 0;
            for (int i = tid; i < M * Nsym; i += NT) if (filt_o) filt_o[i] = 0.f;
        }
        for (int m = 0; m < kMaxTones; m++) sc.f_est[m] = f_est[m];
        if (stats_o && tid == 0) {
            stats_o[0] = f_est[0]; stats_o[1] = f_est[1]; stats_o[2] = f_est[2]; stats_o[3] = f_est[3];
            stats_o[4] = sc.norm_rx_timing; stats_o[5] = sc.SNRest; stats_o[6] = (float)nin; stats_o[7] = sc.ppm;
            stats_o[8] = sc.rx_sig_pow; stats_o[9] = sc.rx_nse_pow;
        }
        pos += (Nmem - nold);
        frame++;
        __syncthreads();
    }

    // ---- save stream state ---------------------------------------------------------------------
    sc.nin = nin;
    for (int i = tid; i < Ndft; i += NT) a.s.Sf[(size_t)sid * Ndft + i] = L.Sf[i];
    if (EXACT0 && a.io.exact0_fmt == PIRIP_KERNEL_WAVE) {
        // The wave kernel's state block (fsk_demod_wave.hip, WaveCfg): the guard area as it stands after a frame -- the frame's last
        // 2 Ts + Ts/4 RAW samples right-aligned in GUARD_B bytes -- then per tone the phase step and oscillator-table row of the frame's
        // tone estimates, then the frame's nin (0 = no frame yet: then nothing is written and the stream stays in its created state)
        if (frame > 0) {
            const int bps = in_u8 ? 2 : in_s16 ? 4 : 8, HIST = 2 * Ts + d.nin_step;
            const int guard_b = ((HIST * bps + 15) / 16) * 16, tail_b = HIST * bps;
            uint32_t *st32 = (uint32_t *)(a.s.hist + (size_t)sid * M * d.hist_len);
            const uint8_t *src = in_base + (size_t)(pos - HIST) * bps;           // (byte copy: a stream's base need not be dword-aligned)
            for (int i = tid; i < tail_b; i += NT) ((uint8_t *)st32)[guard_b - tail_b + i] = src[i];
            // (the guard's padding in front of the tail holds what the wave kernel fills a created stream's guard with: its format's
            //  neutral sample -- capture.hip compares whole state blocks)
            const uint8_t neutral = d.in_format == PIRIP_IN_CU8_FSKDEMOD ? 0x7F : d.in_format == PIRIP_IN_CU8_CSDR ? 0x80 : 0x00;
            for (int i = tid; i < guard_b - tail_b; i += NT) ((uint8_t *)st32)[i] = neutral;
            if (tid == 0) {
                for (int m = 0; m < M; m++) { st32[guard_b / 4 + m] = x0_dth[m]; st32[guard_b / 4 + M + m] = (uint32_t)x0_tix[m]; }
                st32[guard_b / 4 + 2 * M] = (uint32_t)(pos);                 // = nin of the frame (pos counts from 0)
            }
        }
    } else {
        for (int m = 0; m < M; m++)
            for (int h = tid; h < hist_g; h += NT)
                a.s.hist[((size_t)sid * M + m) * d.hist_len + h] = L.hist[m * hist_g + h];
    }
    if (tid == 0) {
        a.s.scal[sid] = sc;
        // (the wave kernel carries no oscillator phase: its handles' theta words stay as reset left them -- capture.hip compares whole states)
        if (!(EXACT0 && a.io.exact0_fmt == PIRIP_KERNEL_WAVE))
            for (int m = 0; m < kMaxTones; m++) a.s.theta[(size_t)sid * kMaxTones + m] = theta[m];
        if (EXACTALL && a.s.phic) for (int m = 0; m < kMaxTones; m++) a.s.phic[(size_t)sid * kMaxTones + m] = phc[m];
        if (EXACT0) { if (a.io.first_out) a.io.first_out[sid] = (int32_t)pos; }
        else {
            if (a.io.nframes) a.io.nframes[sid] = (int32_t)frame;
            if (a.io.consumed) a.io.consumed[sid] = pos;
        }
    }
}

__global__ __launch_bounds__(4 * kWave, 3) void fsk_demod_general_kernel(DemodArgs a) { fsk_demod_general_body<4, 0>(a); }
__global__ __launch_bounds__(8 * kWave, 1) void fsk_demod_general_wide_kernel(DemodArgs a) { fsk_demod_general_body<8, 0>(a); }
__global__ __launch_bounds__(4 * kWave, 1) void fsk_demod_exact0_kernel(DemodArgs a) { fsk_demod_general_body<4, 1>(a); }
__global__ __launch_bounds__(4 * kWave, 1) void fsk_demod_exact_kernel(DemodArgs a) { fsk_demod_general_body<4, 2>(a); }

size_t demod_general_lds_bytes(const FskDims &d) { return kRedBytes + carve(d, nullptr, nullptr); }

hipError_t launch_demod_general(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const size_t lds = demod_general_lds_bytes(a.d);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        // per launch, not cached: the attribute belongs to the current device's copy of the kernel, and handles
        // on several devices / host threads may share this process
        hipError_t e = hipFuncSetAttribute((const void *)fsk_demod_general_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // threads per stream: two waves (measured: 49 G samples/s at config 3 against 37 G with one wave and 36-40 G
    // with four -- barriers and the serial pieces stop scaling) unless the frame is too small to feed them
    // long frames (measured at Ts = 240 / Ndft = 4096, profiles/r03_instance_rates.txt): one stream per CU fits, so its workgroup is wide
    int nt = 2 * kWave;
    if (const char *e = getenv("PIRIP_GENERAL_THREADS")) { const int v = atoi(e); if (v >= 64 && v <= 512 && v % 64 == 0) nt = v; }
    else if (a.d.N + a.d.nin_step < 512) nt = kWave;
    else if (lds > 80 * 1024) nt = 8 * kWave;
    else if (direct_input(a.d)) nt = 4 * kWave;          // long frames, two streams per CU: 45 G at 256 threads against 27 G at 128, 36 G at 512 (one stream per CU)
    if (nt > 4 * kWave) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)fsk_demod_general_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(fsk_demod_general_wide_kernel, dim3(nstreams), dim3(nt), lds, stream, a);
    } else hipLaunchKernelGGL(fsk_demod_general_kernel, dim3(nstreams), dim3(nt), lds, stream, a);
    return hipGetLastError();
}

__global__ void atan2_selftest_kernel(const float *y, const float *x, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = glibc_atan2f(y[i], x[i]);
}
hipError_t selftest_atan2(const float *d_y, const float *d_x, float *d_out, int n)
{
    hipLaunchKernelGGL(atan2_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_y, d_x, d_out, n);
    return hipGetLastError();
}

// The exact first frame exists where the structural tie does: a window of the integrator bank can hold a single sample only when the
// oversample factor equals the samples per symbol (Ts == P: `fsk_demod -p 24` at Ts = 24, the Ts = 8 / 10 shapes) -- there the group size
// of the integrator memory is 1, which the oracle-order window sums need; long frames that are not staged in LDS are left out.
bool demod_exact0_applicable(const FskDims &d) { return d.Ts == d.P && d.grp == 1 && !direct_input(d) && demod_general_lds_bytes(d) <= 160 * 1024; }

// every frame exact (PIRIP_KERNEL=exact): needs single-sample integrator memory (d.grp == 1, set by the handle) and staged input
bool demod_exact_applicable(const FskDims &d) { return d.grp == 1 && !direct_input(d) && demod_general_lds_bytes(d) <= 160 * 1024; }
hipError_t launch_demod_exact(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const size_t lds = demod_general_lds_bytes(a.d);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fsk_demod_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fsk_demod_exact_kernel, dim3(nstreams), dim3(2 * kWave), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_demod_exact0(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const size_t lds = demod_general_lds_bytes(a.d);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fsk_demod_exact0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fsk_demod_exact0_kernel, dim3(nstreams), dim3(2 * kWave), lds, stream, a);
    return hipGetLastError();
}

}  // namespace pirip
