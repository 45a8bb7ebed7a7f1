// pirip_amd/csrc/fsk_demod_general.hip -- general-configuration FSK demodulator kernel (gfx950).
//
// One wavefront (64 lanes) owns one IQ stream and walks its frames in order, because codec2's
// demodulator is frame-serial: nin, the smoothed spectrum Sf, the tone estimates, the local
// oscillator phases and the integrator memory all chain from frame to frame
// [UPSTREAM-RECALLED codec2 fsk.c: fsk_demod_freq_est + fsk_demod_core; SURVEY.md 8a rows
//  a-1, a-4 ... a-8]. Parallelism comes from the batch of independent streams (one workgroup
// each) and from the 64 lanes inside a frame. All per-frame intermediates (complex samples,
// FFT work array, f_dc, f_int) live in LDS; HBM sees the u8/s16 IQ stream once and the bits.
//
// This kernel handles every configuration fsk_create_hbr() accepts (M in {2,4}, any Ts/P/Nsym,
// power-of-two Ndft, peak or mask estimator, four input formats). The specialised kernel in
// fsk_demod_fast.hip overtakes it for the headline configuration; this one stays as the
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

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// arg-max with codec2's tie rule (first maximum wins, only values > 0 count)
__device__ __forceinline__ void wave_argmax(float &v, int &idx)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, kWave);
        int oi = __shfl_xor(idx, o, kWave);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

struct Lds {
    float2 *in;      // [nin_max]
    float2 *X;       // [Ndft]
    float *Sf;       // [Ndft]
    float *Sfw;      // [Ndft]
    float2 *tw;      // [Ndft]
    float *hann;     // [Ndft]
    float *lut;      // [256]
    float2 *fdc;     // [M][Nmem]
    float2 *fint;    // [M][nint]
    uint16_t *perm;  // [Ndft]
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline size_t carve(const FskDims &d, Lds *l, char *base)
{
    size_t off = 0;
    const int nin_max = d.N + d.Ts / 4;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    size_t o_in = take(sizeof(float2) * nin_max);
    size_t o_X = take(sizeof(float2) * d.Ndft);
    size_t o_Sf = take(sizeof(float) * d.Ndft);
    size_t o_Sfw = take(sizeof(float) * d.Ndft);
    size_t o_tw = take(sizeof(float2) * d.Ndft);
    size_t o_hann = take(sizeof(float) * d.Ndft);
    size_t o_lut = take(sizeof(float) * 256);
    size_t o_fdc = take(sizeof(float2) * d.M * d.Nmem);
    size_t o_fint = take(sizeof(float2) * d.M * d.nint);
    size_t o_perm = take(sizeof(uint16_t) * d.Ndft);
    if (l) {
        l->in = (float2 *)(base + o_in); l->X = (float2 *)(base + o_X);
        l->Sf = (float *)(base + o_Sf); l->Sfw = (float *)(base + o_Sfw);
        l->tw = (float2 *)(base + o_tw); l->hann = (float *)(base + o_hann);
        l->lut = (float *)(base + o_lut); l->fdc = (float2 *)(base + o_fdc);
        l->fint = (float2 *)(base + o_fint); l->perm = (uint16_t *)(base + o_perm);
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

}  // namespace

__global__ __launch_bounds__(kWave) void fsk_demod_general_kernel(DemodArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FskDims &d = a.d;
    Lds L;
    carve(d, &L, smem);

    const int tid = threadIdx.x;
    const int sid = blockIdx.x;
    const int M = d.M, Ndft = d.Ndft, Nmem = d.Nmem, nint = d.nint, Ts = d.Ts, P = d.P, Nsym = d.Nsym;
    const int log2n = 31 - __clz(Ndft);

    // ---- load tables and stream state into LDS -------------------------------------------
    for (int i = tid; i < Ndft; i += kWave) {
        L.tw[i] = a.t.tw[i];
        L.hann[i] = a.t.hann[i];
        L.perm[i] = a.t.perm[i];
        L.Sf[i] = a.s.Sf[(size_t)sid * Ndft + i];
    }
    for (int i = tid; i < 256; i += kWave) L.lut[i] = a.t.lut[i];
    for (int i = tid; i < M * Nmem; i += kWave) L.fdc[i] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int m = 0; m < M; m++)
        for (int h = tid; h < d.hist_len; h += kWave)
            L.fdc[m * Nmem + Nmem - d.hist_len + h] = a.s.hist[((size_t)sid * M + m) * d.hist_len + h];

    StreamScalars sc = a.s.scal[sid];
    uint32_t theta[kMaxTones];
#pragma unroll
    for (int m = 0; m < kMaxTones; m++) theta[m] = a.s.theta[(size_t)sid * kMaxTones + m];
    __syncthreads();

    const uint8_t *in_base = a.io.in + (size_t)sid * a.io.in_stride;
    int64_t pos = 0;
    int64_t frame = 0;
    int nin = sc.nin;

    while (frame < a.io.max_frames && pos + nin <= a.io.nsamp) {
        // ---- a-1: convert nin samples to complex float -----------------------------------
        if (d.in_format == PIRIP_IN_CU8_FSKDEMOD || d.in_format == PIRIP_IN_CU8_CSDR) {
            const uint8_t *p = in_base + 2 * pos;
            for (int i = tid; i < nin; i += kWave)
                L.in[i] = make_float2(L.lut[p[2 * i]], L.lut[p[2 * i + 1]]);
        } else if (d.in_format == PIRIP_IN_CS16) {
            const int16_t *p = (const int16_t *)in_base + 2 * pos;
            for (int i = tid; i < nin; i += kWave)
                L.in[i] = make_float2((float)p[2 * i] / (float)PIRIP_FDMDV_SCALE,
                                      (float)p[2 * i + 1] / (float)PIRIP_FDMDV_SCALE);
        } else {
            const float2 *p = (const float2 *)in_base + pos;
            for (int i = tid; i < nin; i += kWave) L.in[i] = p[i];
        }
        __syncthreads();

        // ---- a-5: frequency estimator ------------------------------------------------------
        const int numffts = nin / (Ndft / 2) - 1;
        for (int j = 0; j < numffts; j++) {
            const int off = j * Ndft / 2;
            for (int n = tid; n < Ndft; n += kWave) {
                const int src = L.perm[n];
                const float h = L.hann[src];
                const float2 x = L.in[off + src];
                L.X[n] = make_float2(h * x.x, h * x.y);
            }
            __syncthreads();
            for (int s = 0; s < d.nstages; s++) {
                const int p = a.stages[s].radix, m = a.stages[s].m, fs = a.stages[s].fstride;
                const int nb = Ndft / p;
                for (int b = tid; b < nb; b += kWave) {
                    const int g = b / m, k = b - g * m;
                    float2 *F = L.X + g * p * m + k;
                    if (p == 4) {
                        const float2 t1 = L.tw[k * fs], t2 = L.tw[2 * k * fs], t3 = L.tw[3 * k * fs];
                        float2 f0 = F[0], f1 = F[m], f2 = F[2 * m], f3 = F[3 * m];
                        float2 s0, s1, s2, s3, s4, s5;
                        s0.x = f1.x * t1.x - f1.y * t1.y; s0.y = f1.x * t1.y + f1.y * t1.x;
                        s1.x = f2.x * t2.x - f2.y * t2.y; s1.y = f2.x * t2.y + f2.y * t2.x;
                        s2.x = f3.x * t3.x - f3.y * t3.y; s2.y = f3.x * t3.y + f3.y * t3.x;
                        s5.x = f0.x - s1.x; s5.y = f0.y - s1.y;
                        f0.x += s1.x; f0.y += s1.y;
                        s3.x = s0.x + s2.x; s3.y = s0.y + s2.y;
                        s4.x = s0.x - s2.x; s4.y = s0.y - s2.y;
                        f2.x = f0.x - s3.x; f2.y = f0.y - s3.y;
                        f0.x += s3.x; f0.y += s3.y;
                        f1.x = s5.x + s4.y; f1.y = s5.y - s4.x;
                        f3.x = s5.x - s4.y; f3.y = s5.y + s4.x;
                        F[0] = f0; F[m] = f1; F[2 * m] = f2; F[3 * m] = f3;
                    } else {
                        const float2 t1 = L.tw[k * fs];
                        float2 f0 = F[0], f1 = F[m], t;
                        t.x = f1.x * t1.x - f1.y * t1.y; t.y = f1.x * t1.y + f1.y * t1.x;
                        f1.x = f0.x - t.x; f1.y = f0.y - t.y;
                        f0.x += t.x; f0.y += t.y;
                        F[0] = f0; F[m] = f1;
                    }
                }
                __syncthreads();
            }
            // fftshift + |X| + first-order smoothing
            for (int i = tid; i < Ndft; i += kWave) {
                const float2 x = L.X[(i + Ndft / 2) & (Ndft - 1)];
                const float mag2 = (x.x * x.x) + (x.y * x.y);
                L.Sf[i] = (L.Sf[i] * d.one_minus_tc) + (sqrtf(mag2) * d.tc);
            }
            __syncthreads();
        }

        // peak method (always run: f_est is reported even when the mask method drives the demod)
        int freqi[kMaxTones];
        for (int i = tid; i < Ndft; i += kWave) L.Sfw[i] = L.Sf[i];
        __syncthreads();
        for (int m = 0; m < M; m++) {
            float best = 0.0f; int ib = 0;
            for (int j = d.est_st + tid; j < d.est_en; j += kWave) {
                const float v = L.Sfw[j];
                if (v > best) { best = v; ib = j; }
            }
            wave_argmax(best, ib);
            int f_min = ib - d.f_zero; f_min = f_min < 0 ? 0 : f_min;
            int f_max = ib + d.f_zero; f_max = f_max > Ndft ? Ndft : f_max;
            __syncthreads();
            for (int j = f_min + tid; j < f_max; j += kWave) L.Sfw[j] = 0.0f;
            __syncthreads();
            freqi[m] = ib - Ndft / 2;
        }
        // ascending sort of M <= 4 indices
        for (int x = 1; x < M; x++)
            for (int y = x; y > 0 && freqi[y] < freqi[y - 1]; y--) {
                int t = freqi[y]; freqi[y] = freqi[y - 1]; freqi[y - 1] = t;
            }
        float f_est[kMaxTones] = {0.f, 0.f, 0.f, 0.f};
        uint32_t dtheta[kMaxTones] = {0u, 0u, 0u, 0u};
        int drift_ix[kMaxTones] = {0, 0, 0, 0};
        for (int m = 0; m < M; m++) {
            f_est[m] = (float)freqi[m] * d.bin_hz;
            dtheta[m] = (uint32_t)freqi[m] << (32 - log2n);
            drift_ix[m] = freqi[m] + Ndft / 2;
        }
        if (d.freq_est_type) {
            // mask method: comb of 3-bin teeth slid over Sf, tooth sums in ascending order
            float best = 0.0f; int bb = d.est_st;
            for (int b = d.est_st + tid; b < d.est_en - d.mask_len; b += kWave) {
                float corr = 0.0f;
                for (int k = 0; k < d.n_teeth; k++) corr += L.Sf[b + a.t.teeth[k]];
                if (corr > best) { best = corr; bb = b; }
            }
            // lanes that found nothing keep (0, est_st): smallest index wins ties, as upstream
            wave_argmax(best, bb);
            const float foff = (float)((bb - Ndft / 2) * d.Fs / Ndft);
            const uint32_t base = (uint32_t)(bb - Ndft / 2) << (32 - log2n);
            for (int m = 0; m < M; m++) {
                f_est[m] = foff + (float)(m * d.tone_spacing);
                dtheta[m] = base + a.t.mask_dtheta[m];
                drift_ix[m] = bb * M + m;
            }
        }

        // ---- a-6: shift integrator memory, down-convert, integrate -------------------------
        const int nold = Nmem - nin;
        // shift: the last nold integrator-memory samples move to the front. Source [nin, Nmem) and
        // destination [0, nold) never overlap (nold <= 2.25*Ts < nin), so a direct copy is safe.
        for (int m = 0; m < M; m++)
            for (int i = tid; i < nold; i += kWave) L.fdc[m * Nmem + i] = L.fdc[m * Nmem + nin + i];
        for (int m = 0; m < M; m++) {
            const uint32_t th0 = theta[m], dth = dtheta[m];
            // upstream advances phi_c by a float32-rounded multiplier, so |phi_c| drifts as
            // (1+a)^n inside a frame (renormalised at its end): track that gain to first order
            const float gain_slope = a.t.osc_drift[drift_ix[m]].x;
            for (int j = tid; j < nin; j += kWave) {
                const float2 ph = phasor(th0 + (uint32_t)(j + 1) * dth, L.tw, log2n);
                const float2 x = L.in[j];
                const float g = 1.0f + gain_slope * (float)(j + 1);
                L.fdc[m * Nmem + nold + j] = make_float2((x.x * ph.x + x.y * ph.y) * g, (x.y * ph.x - x.x * ph.y) * g);
            }
            theta[m] = th0 + (uint32_t)nin * dth;
        }
        __syncthreads();
        for (int m = 0; m < M; m++) {
            for (int i = tid; i < nint; i += kWave) {
                const int st = i * Ts / P;
                const float2 *src = L.fdc + m * Nmem + st;
                float2 acc = make_float2(0.f, 0.f);
                for (int j = 0; j < Ts; j++) { acc.x += src[j].x; acc.y += src[j].y; }
                L.fint[m * nint + i] = acc;
            }
        }
        __syncthreads();

        // ---- a-7: fine timing -----------------------------------------------------------------
        float tcr = 0.f, tci = 0.f;
        for (int i = tid; i < nint; i += kWave) {
            float ft1 = 0.f;
            for (int m = 0; m < M; m++) {
                const float2 v = L.fint[m * nint + i];
                ft1 += (v.x * v.x) + (v.y * v.y);
            }
            const float2 ph = a.t.timing_rec[i];   // the upstream recursion's phasor, drift included
            tcr += ft1 * ph.x; tci += ft1 * ph.y;
        }
        tcr = wave_sum(tcr); tci = wave_sum(tci);

        const int frame_bytes = d.pack_bits ? (d.Nbits + 7) / 8 : d.Nbits;
        uint8_t *bits_o = a.io.bits ? a.io.bits + (size_t)sid * a.io.bits_stride + (size_t)frame * frame_bytes : nullptr;
        uint8_t *bits_l = (uint8_t *)L.X;            // FFT work array is free here: staging for packed output (Nbits <= 8*Ndft)
        float *filt_o = a.io.filt ? a.io.filt + (size_t)sid * a.io.filt_stride + (size_t)frame * M * Nsym : nullptr;
        float *stats_o = a.io.stats ? a.io.stats + (size_t)sid * a.io.stats_stride + (size_t)frame * PIRIP_STATS_PER_FRAME : nullptr;

        const bool bad = isnan(tcr) || isnan(tci);
        if (!bad) {
            const float norm_rx_timing = (float)((double)atan2f(tci, tcr) / (2 * M_PI));
            const float rx_timing = norm_rx_timing * (float)P;
            const float d_norm = norm_rx_timing - sc.norm_rx_timing;
            sc.norm_rx_timing = norm_rx_timing;
            if ((double)fabsf(d_norm) < .2) {
                const float appm = (float)(1e6 * d_norm / (float)Nsym);
                sc.ppm = (float)(.9 * sc.ppm + .1 * appm);
            }
            int nin_next = d.N;
            if (!d.burst_mode) {
                if (norm_rx_timing > 0.25f) nin_next = d.N + Ts / 4;
                else if (norm_rx_timing < -0.25f) nin_next = d.N - Ts / 4;
            }

            // ---- a-8: resample, decide, stats ------------------------------------------------
            const int low_sample = (int)floorf(rx_timing);
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = (int)ceilf(rx_timing);
            float sig = 0.f, nse = 0.f, mean_e = 0.f, std_e = 0.f;
            for (int i = tid; i < Nsym; i += kWave) {
                const int st = (i + 1) * P;
                float tmax[kMaxTones];
                float sum = 0.f;
                for (int m = 0; m < M; m++) {
                    const float2 lo = L.fint[m * nint + st + low_sample];
                    const float2 hi = L.fint[m * nint + st + high_sample];
                    float2 t;
                    t.x = (1 - fract) * lo.x; t.y = (1 - fract) * lo.y;
                    t.x = t.x + fract * hi.x; t.y = t.y + fract * hi.y;
                    tmax[m] = (t.x * t.x) + (t.y * t.y);
                    sum += tmax[m];
                }
                float mx = tmax[0]; int sym = 0;
                for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
                if (bits_o) {
                    uint8_t *bo = d.pack_bits ? bits_l : bits_o;
                    if (M == 2) bo[i] = sym == 1;
                    else { bo[2 * i + 1] = sym & 1; bo[2 * i] = (sym & 2) >> 1; }
                }
                if (filt_o) for (int m = 0; m < M; m++) filt_o[m * Nsym + i] = sqrtf(tmax[m]);
                sig += mx;
                nse += (sum - mx) / (float)(M - 1);
                std_e += mx;
                mean_e += sqrtf(mx);
            }
            if (bits_o && d.pack_bits) {
                __syncthreads();
                for (int j = tid; j < frame_bytes; j += kWave) {
                    unsigned byte = 0;
                    for (int b = 0; b < 8; b++) if (8 * j + b < d.Nbits) byte |= (unsigned)(bits_l[8 * j + b] & 1) << (7 - b);
                    bits_o[j] = (uint8_t)byte;
                }
            }
            sig = wave_sum(sig); nse = wave_sum(nse) + 1e-12f;
            mean_e = wave_sum(mean_e); std_e = wave_sum(std_e);
            sig = sig / (float)Nsym; nse = nse / (float)Nsym;
            sc.v_est = (float)sqrt((double)(sig - nse));
            sc.SNRest = sig / nse;
            mean_e = mean_e / (float)Nsym;
            std_e = (std_e / (float)Nsym) - (mean_e * mean_e);
            std_e = std_e > 0.0f ? (float)sqrt((double)std_e) : 0.0f;
            sc.EbNodB = -6 + (20 * log10f((float)((1e-6 + mean_e) / (1e-6 + std_e))));
            sc.snr_est = (float)(.5 * sc.snr_est + .5 * sc.EbNodB);
            nin = nin_next;
        } else {
            // NaN in the timing estimate: upstream returns before touching the outputs
            for (int i = tid; i < frame_bytes; i += kWave) if (bits_o) bits_o[i] = 0;
            for (int i = tid; i < M * Nsym; i += kWave) if (filt_o) filt_o[i] = 0.f;
        }
        for (int m = 0; m < kMaxTones; m++) sc.f_est[m] = f_est[m];
        if (stats_o && tid == 0) {
            stats_o[0] = f_est[0]; stats_o[1] = f_est[1]; stats_o[2] = f_est[2]; stats_o[3] = f_est[3];
            stats_o[4] = sc.norm_rx_timing; stats_o[5] = sc.SNRest; stats_o[6] = (float)nin; stats_o[7] = sc.ppm;
        }
        pos += (Nmem - nold);
        frame++;
        __syncthreads();
    }

    // ---- save stream state ---------------------------------------------------------------------
    sc.nin = nin;
    for (int i = tid; i < Ndft; i += kWave) a.s.Sf[(size_t)sid * Ndft + i] = L.Sf[i];
    for (int m = 0; m < M; m++)
        for (int h = tid; h < d.hist_len; h += kWave)
            a.s.hist[((size_t)sid * M + m) * d.hist_len + h] = L.fdc[m * Nmem + Nmem - d.hist_len + h];
    if (tid == 0) {
        a.s.scal[sid] = sc;
        for (int m = 0; m < kMaxTones; m++) a.s.theta[(size_t)sid * kMaxTones + m] = theta[m];
        if (a.io.nframes) a.io.nframes[sid] = (int32_t)frame;
        if (a.io.consumed) a.io.consumed[sid] = pos;
    }
}

size_t demod_general_lds_bytes(const FskDims &d) { return carve(d, nullptr, nullptr); }

hipError_t launch_demod_general(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const size_t lds = demod_general_lds_bytes(a.d);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)fsk_demod_general_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL(fsk_demod_general_kernel, dim3(nstreams), dim3(kWave), lds, stream, a);
    return hipGetLastError();
}

}  // namespace pirip
