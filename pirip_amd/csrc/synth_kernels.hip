// pirip_amd/csrc/synth_kernels.hip -- synthetic Tx on the device: codec2's fsk_mod
// followed by the u8 IQ quantiser and optional AWGN, one thread per stream, so benchmarks and BER sweeps
// need no host-side modulation or upload.
//
//   fsk_get_test_bits | fsk_mod -c M Fs Rs f1 shift  ->  u8 IQ   (/root/reference/README.md:101,142,218)
//
// The modulator is codec2's continuous-phase recursion [UPSTREAM-RECALLED fsk.c: fsk_mod_c]:
// tx_phase *= dosc[sym] once per sample (float32 complex multiply, no fma), output 2*tx_phase, phase
// renormalised after every block of `norm_syms` symbols (the fsk_mod tool calls fsk_mod_c with
// Nsym = 50 symbols at a time). The recursion is serial per stream; streams are independent, one per
// thread. Noise-free output is bit-identical to the CPU modulator (fsk_plan.cpp FskMod and the oracle).
// Quantiser: u8 = clamp(rintf(127 + amp * x)), float32 arithmetic.
// AWGN: sigma * N(0,1) per component from a counter-based generator (SplitMix64 finaliser of
// (seed, stream, sample) -> two uniforms -> Box-Muller); this is the product's own noise source, it is
// not meant to reproduce any CPU generator.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/pirip_hip.h"

namespace {

struct SynthArgs {
    const uint8_t *bits; size_t bits_stride;     // one bit per byte; stride 0 = every stream sends the same bits
    int64_t nsym;
    uint8_t *out; size_t out_stride; int64_t nsamp;   // samples written per stream (<= nsym*Ts - skip)
    const float *dosc;                           // [nstreams][M][2] per-stream tone phasors exp(j 2 pi f/Fs) (host cosf/sinf)
    const int32_t *skip;                         // [nstreams] leading samples dropped (timing offset), may be null
    int M, Ts, norm_syms, nstreams;
    float amp, sigma;
    uint64_t seed;
};

__device__ __forceinline__ uint64_t splitmix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(64) void synth_kernel(SynthArgs a)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= a.nstreams) return;
    const uint8_t *bits = a.bits + (size_t)s * a.bits_stride;
    uint8_t *out = a.out + (size_t)s * a.out_stride;
    const float *dosc = a.dosc + (size_t)s * a.M * 2;
    float dr[4], di[4];
    for (int m = 0; m < a.M; m++) { dr[m] = dosc[2 * m]; di[m] = dosc[2 * m + 1]; }
    const int bps = a.M == 2 ? 1 : 2;
    const int64_t skip = a.skip ? a.skip[s] : 0;
    float pr = 1.0f, pi = 0.0f;                   // tx_phase_c = comp_exp_j(0)
    int64_t n = -skip;                            // output sample index
    for (int64_t i = 0; i < a.nsym && n < a.nsamp; i++) {
        int sym = 0;
        for (int b = 0; b < bps; b++) sym = (sym << 1) | (bits[i * bps + b] == 1 ? 1 : 0);
        const float cr = dr[sym], ci = di[sym];
        for (int j = 0; j < a.Ts; j++, n++) {
            const float nr = pr * cr - pi * ci;   // cmult(tx_phase_c, dph), -ffp-contract=off
            const float ni = pr * ci + pi * cr;
            pr = nr; pi = ni;
            if (n >= 0 && n < a.nsamp) {
                float xr = 2 * pr, xi = 2 * pi;
                if (a.sigma > 0.f) {
                    const uint64_t r = splitmix(a.seed ^ splitmix(((uint64_t)s << 40) ^ (uint64_t)n));
                    const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);     // (0,1]
                    const float u2 = (float)(uint32_t)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f); // [0,1)
                    const float mag = a.sigma * sqrtf(-2.0f * logf(u1));
                    float sn, cs;
                    sincosf(6.2831853071795865f * u2, &sn, &cs);
                    xr += mag * cs; xi += mag * sn;
                }
                float qr = rintf(127.0f + a.amp * xr), qi = rintf(127.0f + a.amp * xi);
                qr = fminf(fmaxf(qr, 0.f), 255.f); qi = fminf(fmaxf(qi, 0.f), 255.f);
                *(uchar2 *)(out + 2 * n) = make_uchar2((unsigned char)qr, (unsigned char)qi);
            }
        }
        if ((i + 1) % a.norm_syms == 0) {         // comp_normalize() at the end of each fsk_mod_c call
            const float av = sqrtf((pr * pr) + (pi * pi));
            pr = pr / av; pi = pi / av;
        }
    }
}

}  // namespace

extern "C" int pirip_hip_synth_cu8(int Fs, int Rs, int M, int nstreams,
                                   const int32_t *f1_hz, int tone_spacing_hz, const int32_t *skip_samples,
                                   const uint8_t *d_bits, size_t bits_stride, int64_t nsym,
                                   uint8_t *d_out, size_t out_stride_bytes, int64_t nsamp,
                                   float amp, float sigma, uint64_t seed, void *hip_stream)
{
    if (!f1_hz || !d_bits || !d_out || nstreams <= 0 || nsym < 0 || nsamp < 0) return PIRIP_ERR_BAD_ARG;
    if (Fs <= 0 || Rs <= 0 || Fs % Rs || (M != 2 && M != 4) || tone_spacing_hz <= 0) return PIRIP_ERR_BAD_CONFIG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PIRIP_ERR_NO_DEVICE;
    // per-stream tone phasors exactly as fsk_mod computes them: comp_exp_j(2*pi*(f1 + m*spacing)/Fs)
    std::vector<float> dosc((size_t)nstreams * M * 2);
    for (int s = 0; s < nstreams; s++)
        for (int m = 0; m < M; m++) {
            const float w = 2 * M_PI * ((float)(f1_hz[s] + (tone_spacing_hz * m)) / (float)(Fs));
            dosc[((size_t)s * M + m) * 2] = cosf(w);
            dosc[((size_t)s * M + m) * 2 + 1] = sinf(w);
        }
    float *d_dosc = nullptr; int32_t *d_skip = nullptr;
    hipStream_t st = (hipStream_t)hip_stream;
    bool ok = hipMalloc((void **)&d_dosc, sizeof(float) * dosc.size()) == hipSuccess &&
              hipMemcpyAsync(d_dosc, dosc.data(), sizeof(float) * dosc.size(), hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok && skip_samples) {
        ok = hipMalloc((void **)&d_skip, sizeof(int32_t) * (size_t)nstreams) == hipSuccess &&
             hipMemcpyAsync(d_skip, skip_samples, sizeof(int32_t) * (size_t)nstreams, hipMemcpyHostToDevice, st) == hipSuccess;
    }
    if (ok) {
        SynthArgs a{d_bits, bits_stride, nsym, d_out, out_stride_bytes, nsamp, d_dosc, d_skip, M, Fs / Rs,
                    PIRIP_FSK_DEFAULT_NSYM, nstreams, amp, sigma, seed};
        hipLaunchKernelGGL(synth_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, a);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;   // host tables go out of scope
    }
    if (d_dosc) (void)hipFree(d_dosc);
    if (d_skip) (void)hipFree(d_skip);
    return ok ? PIRIP_OK : PIRIP_ERR_HIP;
}
