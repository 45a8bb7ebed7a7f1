// pirip_amd/csrc/fsk_device.hpp -- device-side argument blocks shared by the kernels and the
// C-ABI layer. gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "fsk_plan.hpp"

namespace pirip {

constexpr int kEyeTraces = 8, kEyePoints = 160;   // MODEM_STATS_ET_MAX, MODEM_STATS_EYE_IND_MAX

// Per-stream scalars that codec2 keeps in struct FSK between fsk_demod() calls.
struct StreamScalars {
    int32_t nin;              // samples the next frame consumes (fsk_nin())
    float norm_rx_timing;
    float ppm;
    float snr_est;            // smoothed EbNodB (MODEM_STATS.snr_est)
    float SNRest;
    float EbNodB;
    float v_est;
    float f_est[kMaxTones];
    float rx_sig_pow;         // rx_sig_pow / rx_nse_pow of struct FSK (observable frames, like SNRest)
    float rx_nse_pow;
};

struct DemodTables {
    const float *hann;          // [Ndft]
    const float2 *tw;           // [Ndft] exp(-j 2 pi i / Ndft)
    const uint16_t *perm;       // [Ndft] inverse leaf permutation: FFT work-array slot fed by input index i
    const float *lut;           // [256]
    const float2 *tph;          // [P]
    const int16_t *teeth;       // [n_teeth]
    const uint32_t *mask_dtheta;// mask method: [Ndft*M] per-sample phase step for comb position b, tone m at b*M + m
    const float2 *osc_drift;    // [Ndft*(mask?M:1)] (gain slope a, phase slope d) of the upstream recursion
    const float2 *osc_step;     // same indexing: the float32-rounded per-sample multiplier (cosf, sinf)
    const float2 *timing_rec;   // [nint] fine-timing phasor as the upstream recursion yields it
    const float *fast_tab;      // [16][48] (Ndft == 256 only)
};

struct DemodState {
    float *Sf;                  // [nstreams][Ndft]
    uint32_t *theta;            // [nstreams][kMaxTones]   oscillator phase, 2^32 = one turn
    float2 *hist;               // [nstreams][M][hist_len] last integrator-memory samples
    StreamScalars *scal;        // [nstreams]
    float2 *phic;               // [nstreams][kMaxTones] the exact kernel's carried oscillators phi_c (0, 0 = created state); nullptr elsewhere
};

// Fused FSK_LDPC hand-over (ldpc_kernels.hip): instead of soft magnitudes the demodulator writes, per frame, the
// Nbits bit log-likelihood ratios and their hard decisions packed 32 per word (first bit in the MSB) straight into the LDPC
// receiver's work buffers -- the magnitudes never travel through HBM.
struct SoftOut {
    uint16_t *llr; size_t llr_stride;       // [stream][bit0 + frame * Nbits + bit], IEEE binary16 (nullptr: not requested)
    uint32_t *words; size_t words_stride;   // [stream][word] over the same bit positions; bit0 % 32 == 0; pre-zeroed by the caller
    const float *lnI0;                      // ln I0(j / 8), j = 0 .. 257
    int bit0;                               // bits of history in front of this call's first frame (2 * bits_per_frame)
    int llr_map;                            // kLlrUpstream / kLlrRician (fsk_ldpc.hpp: the code file's `llr_map` key)
};

// Soft magnitudes -> symbol metric of codec2's fsk_rx_filt_to_llrs() [UPSTREAM-RECALLED mpdecode_core.c: FskDemod + logbesseli0, CML's
// piecewise-quadratic fit of ln I0]: metric = logbesseli0(2 * SNRest * |r| / v_est). Here the frame's factor k = (2 * SNRest) / v_est is
// formed once and the argument is k * |r| (upstream forms sqrt(r^2 / v_est^2) per value: the same number up to the last float bits,
// far below the binary16 rounding of the soft bits); the polynomial in float32, every product and sum rounded once, in this order --
// the LLR stage (ldpc_kernels.hip), the demodulator's fused hand-over (fsk_demod_wave.hip) and the checker restate exactly this.
constexpr int kLlrUpstream = 0, kLlrRician = 1;
constexpr float kLlrMaxUpstream = 1000.0f;  // keeps the binary16 soft bits finite when v_est is tiny (any |LLR| >= 32 saturates the decoder)
__device__ __forceinline__ float logbesseli0_upstream(float x)
{
    float c2 = 0.226f, c1 = 0.0125f, c0 = -0.0012f;
    if (x >= 1.0f) { c2 = 0.1245f; c1 = 0.2177f; c0 = -0.108f; }
    if (x >= 2.0f) { c2 = 0.0288f; c1 = 0.6314f; c0 = -0.5645f; }
    if (x >= 5.0f) { c2 = 0.002f; c1 = 0.9048f; c0 = -1.2997f; }
    if (x >= 20.0f) { c2 = 0.0f; c1 = 0.9867f; c0 = -2.2053f; }
    return (((c2 * x) * x) + (c1 * x)) + c0;
}
// x / C correctly rounded for x = 0 or x >= 2^-125 (C = 3, 50: compared with the IEEE quotient for every such float on the CPU,
// tools/div_const_check.c, and on the device, pirip_hip_selftest_div): q = x * RN(1/C), one residual correction.
template <int C>
__device__ __forceinline__ float div_rn_const(float x)
{
    if constexpr (C == 1) return x;
    constexpr float c = (float)C, rc = 1.0f / c;
    const float q = x * rc;
    return __builtin_fmaf(__builtin_fmaf(-q, c, x), rc, q);
}
// the frame's factor: upstream 2 * (sig / nse) / v_est, Rician 2 * v_est / nse, with v_est = sqrt(sig - nse) (0 when sig <= nse)
__device__ __forceinline__ float llr_frame_gain(int llr_map, float sig, float nse)
{
    const float a2 = sig - nse;
    const float amp = a2 > 0.f ? sqrtf(a2) : 0.f;
    if (llr_map == kLlrRician) return (2.0f * amp) / nse;
    return amp > 0.f ? (2.0f * (sig / nse)) / amp : 0.f;
}

// One long capture run as many segments (capture.hip): stream `sid` of the launch is a segment of the SAME input, with its own first
// sample, frame budget and place in the common output (strides 0). max_frames < 0: this stream slot sits the launch out.
struct SegDesc {
    int64_t in_off;             // first sample of the segment, relative to DemodIO::in
    int64_t out_frame0;         // output frame index of the segment's first frame (bits / rx_filt / stats rows)
    int32_t max_frames;
    int32_t reserved;
};

struct DemodIO {
    const uint8_t *in; size_t in_stride; int64_t nsamp;
    uint8_t *bits; size_t bits_stride;
    float *filt; size_t filt_stride;
    float *stats; size_t stats_stride;
    int32_t *nframes; int64_t *consumed;
    int64_t max_frames;
    SoftOut soft;
    const SegDesc *seg;         // nullptr: every stream starts at its own in + sid * in_stride (the batch entry points)
    float *eye;                 // nullptr, or [nstreams][kEyeTraces][kEyePoints]: |f_int| eye traces of each stream's latest frame (general kernel only)
    // The first frame of a stream after fsk_create / reset, in the oracle's own operation order (fsk_demod_exact0_kernel):
    // a launch of the demodulator proper that follows such a prologue starts each stream at first[sid] samples / one frame in.
    const int32_t *first;       // nullptr, or [nstreams]: samples the prologue consumed for the stream in THIS call (0: none)
    int32_t *first_out;         // the prologue's side of it
    int exact0_fmt;             // prologue only: layout of the state it leaves (PIRIP_KERNEL_GENERAL / _WAVE)
};

struct DemodArgs {
    FskDims d;
    FftStage stages[kMaxStages];
    DemodTables t;
    DemodState s;
    DemodIO io;
    float tw_s2[18];            // wave-uniform FFT twiddles of the wave kernel (Ndft 256: stage 2; Ndft 512: tw[64], tw[128], tw[192])
};

// launchers (fsk_demod_kernels.hip)
size_t demod_general_lds_bytes(const FskDims &d);
hipError_t launch_demod_general(const DemodArgs &a, int nstreams, hipStream_t stream);
// the exact first frame (a variant of the general kernel): applicable when a window can hold a single sample (Ts == P) and the shape fits
bool demod_exact0_applicable(const FskDims &d);
hipError_t launch_demod_exact0(const DemodArgs &a, int nstreams, hipStream_t stream);
// every frame in the oracle's operation order (PIRIP_KERNEL=exact; kernel kind PIRIP_KERNEL_EXACT)
bool demod_exact_applicable(const FskDims &d);
hipError_t launch_demod_exact(const DemodArgs &a, int nstreams, hipStream_t stream);
hipError_t selftest_atan2(const float *d_y, const float *d_x, float *d_out, int n);   // the prologue's atan2f restatement on device arrays
// wave-per-stream kernel (fsk_demod_wave.hip): Ts = 24 / Ndft = 256 and Ts = 40 / Ndft = 512 instances; returns
// hipErrorNotSupported when no instance applies
bool demod_wave_applicable(const FskDims &d);
int demod_wave_describe(const FskDims &d, char *buf, size_t n);   // instance name of the configuration (0 when none applies)
int demod_wave_streams_per_cu(const FskDims &d);                   // streams of the instance resident on one CU (waves per SIMD x 4)
bool demod_wave_soft_capable(const FskDims &d);                    // the instance can write SoftOut (bit LLRs + hard words)
int64_t demod_wave_max_samples(const FskDims &d);
hipError_t launch_demod_wave(const DemodArgs &a, int nstreams, hipStream_t stream);
// workgroup-per-stream kernel for long symbols (fsk_demod_block.hip): Ts = 240 / Ndft = 4096 instances (rtl_fsk -r 1000 at 240 kS/s)
bool demod_block_applicable(const FskDims &d);
int demod_block_describe(const FskDims &d, char *buf, size_t n);
hipError_t launch_demod_block(const DemodArgs &a, int nstreams, hipStream_t stream);
// the launcher of a handle's kernel kind (PIRIP_KERNEL_GENERAL / _WAVE / _BLOCK)
inline hipError_t launch_demod_kind(int kind, const DemodArgs &a, int nstreams, hipStream_t stream)
{
    return kind == 2 ? launch_demod_wave(a, nstreams, stream) : kind == 3 ? launch_demod_block(a, nstreams, stream) :
           kind == 4 ? launch_demod_exact(a, nstreams, stream) : launch_demod_general(a, nstreams, stream);
}
const char *demod_wave_source_hash();                              // Makefile: sha256 prefix of the gfx950 code object in fsk_demod_wave.o
// exhaustive device-side check of the wave kernel's correctly rounded square roots (x = 0 and every float in [2^-96, FLT_MAX])
hipError_t selftest_sqrt(unsigned long long *mismatches);
// the fused FSK_LDPC hand-over's x / 3 and x / 50 (x * RN(1/c) corrected once) against the IEEE quotient: x = 0 and every float in [2^-125, FLT_MAX]
hipError_t selftest_div(unsigned long long *mismatches);       // (count for / 50, saturated) << 32 | count for / 3

}  // namespace pirip
