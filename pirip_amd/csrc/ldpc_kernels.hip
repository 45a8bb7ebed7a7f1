// pirip_amd/csrc/ldpc_kernels.hip -- FSK_LDPC receive on the GPU (include/pirip_hip.h section E):
//   soft decisions (fsk_demod_sd's rx_filt) -> bit LLRs -> 32-bit unique-word search / sync state machine ->
//   sum-product LDPC decode (<= max_iter iterations, parity-check count, iteration count) -> CRC16 -> packed payload
//   bytes + rx_status, one record per demodulator call -- the stream `rtl_fsk --code ... -b` feeds to frame_repeater
//   (/root/reference/tx/frame_repeater.c:55-62,71,80,88; README.md:176-212).
// The parity-check matrix, the unique word and the sync thresholds are run-time DATA (fsk_ldpc.hpp): codec2's
// H_256_512_4 is not in /root/reference, nothing here is specific to a stand-in.
//
// Stages (all streams of a batch at once):
//   llr_tile_kernel  one workgroup per 32 demod calls of a stream: sig/nse of each frame, then Nbits LLRs per call (non-coherent
//                  M-FSK, ln I0 by table + linear interpolation; 4-FSK bits by max-log) and their hard decisions 32 per word
//                  (hard_kernel does the packing when a code's two-frame window is not a whole number of words)
//   uwbest_kernel  best unique-word position (fewest errors, earliest) of every call's search window, from the packed words
//   fsm_kernel     one lane per stream walks its calls in order (the state machine is serial and tiny) and lists the frames
//                  to decode
//   decode_kernel  one wave per listed frame, eight waves per workgroup walking their stream's list with H staged once:
//                  flooding sum-product in the phi domain, H / phi table / messages in LDS, each lane's check rows in registers
// Every floating-point step is written so that the CPU oracle can mirror it operation for operation (table look-ups,
// fixed summation order, no fused multiply-add: the file is built with -ffp-contract=off): hard outputs are bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pirip_hip.h"
#include "fsk_device.hpp"
#include "fsk_ldpc.hpp"

using namespace pirip;

// pirip_capi.hip: the demodulator's side of the fused hand-over
namespace pirip {
int demod_batch_soft(pirip_hip_demod *h, const void *d_in, size_t in_stride_bytes, int64_t nsamp, const SoftOut &so, float *d_stats, size_t stats_stride,
                     int32_t *d_nframes, int64_t *d_consumed, int64_t max_frames, hipStream_t st, int s0 = 0, int n = -1);
int demod_handle_shape(const pirip_hip_demod *h, int *M, int *Nsym, int *nstreams, int *device);
int demod_streams_per_cu(const pirip_hip_demod *h);                     // streams of the handle's wave instance that one CU holds at a time (0: another kernel)
bool demod_soft_capable(const pirip_hip_demod *h, int64_t nsamp);      // the handle's kernel instance can write the fused hand-over for calls of nsamp samples
}

namespace {

constexpr int kWave = 64;
constexpr int kLnI0N = 256;              // ln I0 table: x = j/8, j = 0..256
// phi(x) = -ln tanh(x/2) over the range codec2's phi0() covers [UPSTREAM-RECALLED mpdecode_core.c / phi0.c, CML's MpDecode: "if (x > 10)
// return 0; else if (x < 9.08e-5) return 10; ..." -- the two end clamps are recalled, the staircase in between is not and is replaced
// by the function itself, 32 bins per octave]: messages saturate at 10 and vanish beyond 10. Rounds 2-4 ran [2^-24, 32) (saturation at
// 17.3); tools/ldpc_precision.py shows that this range, not the binary16 soft bits or the table's resolution, is what moves frames
// across the decoding edge against a double-precision receiver.
constexpr int kPhiLoExp = -14, kPhiHiExp = 4, kPhiSteps = 32;
constexpr int kPhiN = (kPhiHiExp - kPhiLoExp) * kPhiSteps;   // 576 bins, 32 per octave, x in [2^-14, 16)
constexpr float kPhiXLo = 9.08e-5f, kPhiXHi = 10.0f;        // below: phi = 10; from kPhiXHi up: phi = 0 (the table's bins from 10.0 on hold 0)
constexpr float kLlrMax = 24.0f;
constexpr int kDegFast = 8;
// "register-resident rows" decoder variant: each lane keeps the column lists of its <= kRowsPerLane check rows in VGPRs (packed
// u16) for the whole workgroup's life -- the code is fixed per handle, so these LDS reads would otherwise repeat in the check
// and parity passes of every iteration of every frame. (Doing the same for the variable-node edge lists spills at 128 VGPRs.)
constexpr int kRowsPerLane = 4;
constexpr int kInfoPerCall = PIRIP_LDPC_INFO_PER_CALL;   // state, uw_loc, uw_err, bad_uw, iter, pcc, decoded frame's window position (-1 none), crc_ok, eraw, 0

struct LdpcDev {
    int n, k, m, E, max_iter, uw_thresh1, uw_thresh2, bad_uw_thresh, M, Nsym, Nbits, bpf;
    int max_row_deg;                     // largest check-node degree (rows up to kDegFast keep their phi terms in registers)
    uint32_t uw_word;                    // unique word, first bit in the MSB
    const uint16_t *row_ptr, *col_idx, *col_ptr, *col_edge;
    const float *lnI0, *phi;
    int llr_map;                         // kLlrUpstream (codec2's fsk_rx_filt_to_llrs as recalled, the default) / kLlrRician (code file key `llr_map`)
};

struct FsmState { int32_t state, loc, bad_uw, uw_err; };

// Soft bits are exchanged as IEEE binary16, round to nearest even (ldpc_oracle.c, the checker, says why): h16 is the storage type.
typedef uint16_t h16;
__device__ __forceinline__ h16 f2h(float x) { return __builtin_bit_cast(h16, (_Float16)x); }
__device__ __forceinline__ float h2f(h16 u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ float round16(float x) { return (float)(_Float16)x; }
template <typename OUT> __device__ __forceinline__ OUT to_out(float rounded);
template <> __device__ __forceinline__ h16 to_out<h16>(float rounded) { return f2h(rounded); }      // exact: the value is a binary16 already
template <> __device__ __forceinline__ float to_out<float>(float rounded) { return rounded; }

// ln I0(x), x >= 0: table at multiples of 1/8 up to 32 with linear interpolation, slope 1 beyond
// (branch-free: beyond 32 the argument is held at 32, where the interpolation gives tab[256] + 0 exactly, and x - 32 is added)
__device__ __forceinline__ float ln_i0(const float *tab, float x)
{
    const bool in = x < 32.0f;
    const float xs = (in ? x : 32.0f) * 8.0f;
    const int j = (int)xs;
    const float f = xs - (float)j;
    const float t0 = tab[j], t1 = tab[j + 1];
    return (t0 + (f * (t1 - t0))) + (in ? 0.0f : x - 32.0f);
}

// phi(x) = -ln tanh(x/2) by bins of the float's exponent and top five mantissa bits; x is clamped to [9.08e-5, 10]
__device__ __forceinline__ float phi_lookup(const float *tab, float x)
{
    const float lo = kPhiXLo;
    if (!(x >= lo)) x = lo;
    if (x >= kPhiXHi) return 0.0f;
    const int idx = (int)(__builtin_bit_cast(uint32_t, x) >> 18) - (int)((uint32_t)(127 + kPhiLoExp) << 5);
    return tab[idx];
}

// Sum over a wave in the receiver's DEFINED order (the checker (ldpc_oracle.c): wave_order_sum): row_shr 1, 2, 4, 8 inside rows of 16 lanes,
// then row 1 += row 0's total and row 3 += row 2's, then rows 2 and 3 += lane 31's; lane 63 holds the result. The same DPP steps as
// the demodulator's wsum() -- that is the point: the frame's signal / noise sums of the LLR stage cost the fused hand-over 14
// instructions instead of 100 dependent adds. Terms are >= 0 (adding the +0 of an absent source lane changes nothing).
#define LDPC_DPP_F(src, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(src)), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_order_sum(float v)
{
    v = v + LDPC_DPP_F(v, 0x111, 0xf);
    v = v + LDPC_DPP_F(v, 0x112, 0xf);
    v = v + LDPC_DPP_F(v, 0x114, 0xf);
    v = v + LDPC_DPP_F(v, 0x118, 0xf);
    v = v + LDPC_DPP_F(v, 0x142, 0xa);
    v = v + LDPC_DPP_F(v, 0x143, 0xc);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef LDPC_DPP_F

// ---- stage 1: LLRs ----------------------------------------------------------------------------------------------------
// grid (ceil(ncalls / kLlrTile), nstreams), block 256: a workgroup turns kLlrTile consecutive demodulator calls of one stream
// (a contiguous run of rx_filt, read as rows of Nsym consecutive floats) into soft bits. llr_all[s] = [2*bpf history | ncalls*Nbits
// new]; tile 0 also brings the history in. The frame statistics (codec2's fsk_demod_core: sig and nse, sums over the symbols) are
// summed in the receiver's defined wave order (wave_order_sum above; the oracle states the same order in C): cheap here and in the
// demodulator's fused hand-over, and a last-bit difference from a serial sum vanishes in the binary16 rounding of the soft bits.
// When `words` is given the tile also packs its hard decisions 32 per word (first bit in the MSB) -- it covers whole words
// because the host only asks for that when 2*bpf is a multiple of 32 (kLlrTile * Nbits always is).
constexpr int kLlrTile = 32;
constexpr int kLlrThreads = 256;

// OUT: h16 inside the receiver; float (the same values, widened) for the stand-alone pirip_hip_ldpc_llr entry
template <bool REG, typename OUT>     // REG: Nsym <= 64, a call's magnitudes are read once into registers (one lane per symbol) and serve both passes
__global__ __launch_bounds__(kLlrThreads) void llr_tile_kernel(LdpcDev c, const float *rx_filt, size_t filt_stride, const int32_t *ncalls_s,
                                                               int ncalls, OUT *llr_all, size_t llr_stride, const h16 *llr_hist,
                                                               uint32_t *words, int nwords)
{
    extern __shared__ __attribute__((aligned(16))) float sm_llr[];
    const int per = c.M * c.Nsym;                          // magnitudes per call, fsk_demod_sd layout [m][sym]: read from global memory
                                                           // (rows of Nsym consecutive floats per lane group; the second pass hits L2)
    float *s_t = sm_llr;                                   // [tile][Nsym][2] (max |.|^2, noise term), then [tile][2 Nsym] soft bits
    float *s_g = s_t + kLlrTile * 2 * c.Nsym;              // [tile] 2 A / sigma^2
    float *s_sn = s_g + kLlrTile;                          // [tile][2] the calls' signal / noise sums, until the gains are formed
    float *s_i0 = s_sn + 2 * kLlrTile;                     // [kLnI0N + 2] (16-byte aligned: the tile sizes are multiples of 4); upstream mapping: 5 rows (c2, c1, c0, -) instead
    const int tid = threadIdx.x, s = blockIdx.y;
    const int call0 = blockIdx.x * kLlrTile;
    const int ncl = (ncalls - call0) < kLlrTile ? (ncalls - call0) : kLlrTile;
    const int valid = ncalls_s ? ncalls_s[s] : ncalls;
    OUT *dst = llr_all + (size_t)s * llr_stride;
    uint32_t *wdst = words ? words + (size_t)s * nwords : nullptr;

    if (c.llr_map == kLlrRician) { for (int i = tid; i <= kLnI0N + 1; i += kLlrThreads) s_i0[i] = c.lnI0[i]; }
    else if (tid < 20) {
        // codec2's logbesseli0 pieces (fsk_device.hpp: logbesseli0_upstream) as rows of LDS, picked per value by segment index: as
        // selects they are twelve v_cndmask per value (the demodulator's fused hand-over keeps the same rows)
        const int sg = tid >> 2, cc = tid & 3;
        const float c2 = sg == 0 ? 0.226f : sg == 1 ? 0.1245f : sg == 2 ? 0.0288f : sg == 3 ? 0.002f : 0.0f;
        const float c1 = sg == 0 ? 0.0125f : sg == 1 ? 0.2177f : sg == 2 ? 0.6314f : sg == 3 ? 0.9048f : 0.9867f;
        const float c0 = sg == 0 ? -0.0012f : sg == 1 ? -0.108f : sg == 2 ? -0.5645f : sg == 3 ? -1.2997f : -2.2053f;
        s_i0[tid] = cc == 0 ? c2 : cc == 1 ? c1 : cc == 2 ? c0 : 0.0f;
    }
    if (blockIdx.x == 0 && llr_hist) {
        const h16 *hs = llr_hist + (size_t)s * 2 * c.bpf;
        for (int i = tid; i < 2 * c.bpf; i += kLlrThreads) dst[i] = to_out<OUT>(h2f(hs[i]));
        if (wdst)
            for (int w = tid; w < (2 * c.bpf) / 32; w += kLlrThreads) {
                uint32_t v = 0;
                for (int b = 0; b < 32; b++) if (h2f(hs[32 * w + b]) < 0.0f) v |= 0x80000000u >> b;
                wdst[w] = v;
            }
    }
    // (loops are wave-per-call, lane-per-element: no integer division by the run-time frame sizes in any inner loop)
    const int lane = tid & (kWave - 1), wv = tid >> 6;
    constexpr int kWaves = kLlrThreads / kWave;
    const float *src = rx_filt + (size_t)s * filt_stride + (size_t)call0 * per;
    constexpr int kPerWave = kLlrTile / kWaves;
    float vreg[REG ? kPerWave : 1][4];
    if constexpr (REG) {
        // all of the wave's loads go out together: 8 calls x M tones, one symbol per lane
#pragma unroll
        for (int q = 0; q < kPerWave; q++) {
            const int cl = wv + kWaves * q;
            const bool live = cl < ncl && call0 + cl < valid && lane < c.Nsym;
#pragma unroll
            for (int m = 0; m < 4; m++) vreg[q][m] = (live && m < c.M) ? src[(size_t)cl * per + m * c.Nsym + lane] : 0.0f;
        }
    }
    // per (call, symbol): the largest tone power and the mean of the others (codec2's per-symbol terms); the frame's two sums in
    // wave order: lane l adds its symbols l, l + 64, ... in index order, then the lanes combine (wave_order_sum)
    // (the sums of a call are wave-uniform; the gain -- two divisions and a square root -- is formed afterwards, one call per lane, instead of
    //  by all 64 lanes once per call)
    auto frame_gain = [&](int cl, float sig_l, float nse_l) {
        const float sig = wave_order_sum(sig_l), nse = wave_order_sum(nse_l);
        if (lane == 0) { s_sn[2 * cl] = sig; s_sn[2 * cl + 1] = nse; }
    };
    if constexpr (REG) {
#pragma unroll
        for (int q = 0; q < kPerWave; q++) {
            const int cl = wv + kWaves * q;
            float sum = 0.f, mx = 0.f;
#pragma unroll
            for (int m = 0; m < 4; m++) if (m < c.M) { const float p = vreg[q][m] * vreg[q][m]; sum = sum + p; mx = p > mx ? p : mx; }
            const bool on = cl < ncl && lane < c.Nsym;                                     // (vreg is zero elsewhere, the terms too)
            // (x / 3 as x * RN(1/3) corrected once: the IEEE quotient for every finite x >= 0 -- fsk_device.hpp: div_rn_const; wave-uniform test for the rest)
            const float oth = sum - mx;
            float mean_oth;
            if (c.M == 4 && __all(!(oth > 3.0e38f))) mean_oth = div_rn_const<3>(oth);
            else mean_oth = oth / (float)(c.M - 1);
            if (cl < ncl) frame_gain(cl, on ? mx : 0.0f, on ? mean_oth : 0.0f);
        }
    } else {
        for (int cl = wv; cl < ncl; cl += kWaves) {
            const bool live = call0 + cl < valid;
            float sig_l = 0.f, nse_l = 0.f;
            for (int i = lane; i < c.Nsym; i += kWave) {
                float sum = 0.f, mx = 0.f;
                for (int m = 0; m < c.M; m++) { const float v = live ? src[(size_t)cl * per + m * c.Nsym + i] : 0.0f; const float p = v * v; sum = sum + p; mx = p > mx ? p : mx; }
                sig_l = sig_l + mx;
                nse_l = nse_l + ((sum - mx) / (float)(c.M - 1));
            }
            frame_gain(cl, sig_l, nse_l);
        }
    }
    __syncthreads();
    if (tid < ncl) {
        const float sig = s_sn[2 * tid] / (float)c.Nsym;
        const float nse = (s_sn[2 * tid + 1] / (float)c.Nsym) + 1e-12f;
        s_g[tid] = llr_frame_gain(c.llr_map, sig, nse);
    }
    __syncthreads();
    const int bps = c.M == 2 ? 1 : 2;
    auto soft_bits = [&](int cl, int i, const float *mag) {
        const float g = s_g[cl];
        float L[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; m++) if (m < c.M) {
            const float x = g * mag[m];
            if (c.llr_map == kLlrRician) L[m] = ln_i0(s_i0, x);
            else {
                int sg = x >= 1.0f ? 1 : 0;                                  // (a chain of selects: 2 instructions per threshold, as a sum 3)
                sg = x >= 2.0f ? 2 : sg; sg = x >= 5.0f ? 3 : sg; sg = x >= 20.0f ? 4 : sg;
                const float4 cf = ((const float4 *)s_i0)[sg];
                L[m] = (((cf.x * x) * x) + (cf.y * x)) + cf.z;       // = logbesseli0_upstream(x), operation for operation
            }
        }
        // Somap with max_star0 = max and the sign flip: bit LLR = best metric among the symbols whose bit is 0 - best among those whose bit is 1
        float l0, l1 = 0.f;
        if (c.M == 2) l0 = L[0] - L[1];
        else {
            l0 = (L[0] > L[1] ? L[0] : L[1]) - (L[2] > L[3] ? L[2] : L[3]);      // MSB: symbols 0,1 vs 2,3
            l1 = (L[0] > L[2] ? L[0] : L[2]) - (L[1] > L[3] ? L[1] : L[3]);      // LSB: symbols 0,2 vs 1,3
        }
        const float lmax = c.llr_map == kLlrRician ? kLlrMax : kLlrMaxUpstream;
        l0 = l0 > lmax ? lmax : (l0 < -lmax ? -lmax : l0);
        l1 = l1 > lmax ? lmax : (l1 < -lmax ? -lmax : l1);
        const bool live = call0 + cl < valid;                                    // no demodulator output for this call: neutral soft bits
        s_t[cl * 2 * c.Nsym + bps * i] = live ? round16(l0) : 0.0f;          // what is handed over is the binary16 value: signs below follow it
        if (bps == 2) s_t[cl * 2 * c.Nsym + 2 * i + 1] = live ? round16(l1) : 0.0f;
    };
    if constexpr (REG) {
#pragma unroll
        for (int q = 0; q < kPerWave; q++) {
            const int cl = wv + kWaves * q;
            if (cl < ncl && lane < c.Nsym) soft_bits(cl, lane, vreg[q]);
        }
    } else {
        for (int cl = wv; cl < ncl; cl += kWaves)
            for (int i = lane; i < c.Nsym; i += kWave) {
                const bool live0 = call0 + cl < valid;
                float mag[4] = {0.f, 0.f, 0.f, 0.f};
                for (int m = 0; m < c.M; m++) mag[m] = live0 ? src[(size_t)cl * per + m * c.Nsym + i] : 0.0f;
                soft_bits(cl, i, mag);
            }
    }
    __syncthreads();
    OUT *out = dst + 2 * c.bpf + (size_t)call0 * c.Nbits;
    const int nb = ncl * c.Nbits;
    // the tile's soft bits in stream order, 64 per wave step: bit i of the tile is bit i - cl Nbits of call cl = i / Nbits (exact as
    // mulhi(i, ceil(2^32 / Nbits)) for i < 2^16); the hard decisions of a step are one ballot = two words, first bit in the MSB
    const uint32_t magic = (uint32_t)((((uint64_t)1 << 32) + (uint32_t)c.Nbits - 1u) / (uint32_t)c.Nbits);
    const int w0 = (2 * c.bpf + call0 * c.Nbits) / 32;
    for (int i0 = 64 * wv; i0 < nb; i0 += 64 * kWaves) {
        const int i = i0 + lane;
        const int cl = (int)__umulhi((uint32_t)i, magic), b = i - cl * c.Nbits;
        const float v = i < nb ? s_t[cl * 2 * c.Nsym + b] : 0.0f;
        if (i < nb) out[i] = to_out<OUT>(v);
        if (wdst) {
            const unsigned long long neg = __ballot(v < 0.0f);
            if (lane < 2 && i0 + 32 * lane < nb) wdst[w0 + (i0 >> 5) + lane] = __builtin_bitreverse32((uint32_t)(neg >> (32 * lane)));
        }
    }
    if (wdst && blockIdx.x == gridDim.x - 1)                                     // the last tile also writes the zero tail
        for (int w = (nb + 31) / 32 + tid; w < nwords - w0; w += kLlrThreads) wdst[w0 + w] = 0;
}

size_t llr_tile_lds(const LdpcDev &c) { return sizeof(float) * ((size_t)kLlrTile * 2 * c.Nsym + 3 * kLlrTile + kLnI0N + 2); }

template <typename OUT>
hipError_t launch_llr(const LdpcDev &c, dim3 grid, hipStream_t st, const float *rx_filt, size_t filt_stride, const int32_t *ncalls_s, int ncalls,
                      OUT *llr_all, size_t llr_stride, const h16 *llr_hist, uint32_t *words, int nwords)
{
    const size_t lds = llr_tile_lds(c);
    if (c.Nsym <= kWave) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)llr_tile_kernel<true, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((llr_tile_kernel<true, OUT>), grid, dim3(kLlrThreads), lds, st, c, rx_filt, filt_stride, ncalls_s, ncalls, llr_all, llr_stride, llr_hist, words, nwords);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)llr_tile_kernel<false, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((llr_tile_kernel<false, OUT>), grid, dim3(kLlrThreads), lds, st, c, rx_filt, filt_stride, ncalls_s, ncalls, llr_all, llr_stride, llr_hist, words, nwords);
    }
    return hipGetLastError();
}

// hard decisions, 32 per word, first bit in the MSB; words[s][w] covers llr_all[s][32 w .. 32 w + 32) (zero beyond the end)
__global__ void hard_kernel(const h16 *llr_all, size_t llr_stride, int nbits_total, uint32_t *words, int nwords)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (w >= nwords) return;
    const h16 *src = llr_all + (size_t)s * llr_stride;
    uint32_t v = 0;
    for (int b = 0; b < 32; b++) { const int i = 32 * w + b; if (i < nbits_total && h2f(src[i]) < 0.0f) v |= 0x80000000u >> b; }
    words[(size_t)s * nwords + w] = v;
}

__device__ __forceinline__ uint32_t window32(const uint32_t *words, int p)
{
    const uint32_t a = words[p >> 5], b = words[(p >> 5) + 1];
    const int sh = p & 31;
    return sh ? ((a << sh) | (b >> (32 - sh))) : a;
}

// unique-word errors at bit position p of a stream (window [p, p+32) inside its bits; 255 where the window does not fit)
__device__ __forceinline__ int uw_errors(const uint32_t *words, int p, int nbits_total, uint32_t uw)
{
    return p + 32 <= nbits_total ? __popc(window32(words, p) ^ uw) : 255;
}

// best unique-word position of every call's search window: key = (errors << 16) | position, minimised (fewest errors, then the
// earliest position -- the serial scan's first minimum). The windows of consecutive calls overlap (bpf positions each, Nbits
// apart), so a position's error count -- a funnel shift, an xor and a popcount on the hard-decision words staged in LDS -- is
// computed ONCE: positions are cut into chunks of Nbits aligned with the calls, sixteen lanes reduce a chunk to two keys (the
// minimum over all of it and over its first bpf % Nbits positions), and call c's window is chunks c .. c+K-1 whole plus the head
// of chunk c+K, K = bpf / Nbits (round 5; before, every call scanned its own bpf positions: 5.4 evaluations per position for the
// 4-FSK shape). The state machine below reads the key when it is searching instead of scanning bpf positions itself.
constexpr int kUwCalls = 56, kUwLanes = 16;                  // calls per workgroup; lanes per chunk
__global__ __launch_bounds__(256) void uwbest_kernel(LdpcDev c, int ncalls, const uint32_t *words, int nwords, int nbits_total, uint32_t *best)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_w[];
    const int s = blockIdx.y;
    const int K = c.bpf / c.Nbits, rem = c.bpf - K * c.Nbits;
    const int call0 = blockIdx.x * kUwCalls;
    const int ncl = (ncalls - call0) < kUwCalls ? (ncalls - call0) : kUwCalls;
    const int nch = ncl + K - (rem ? 0 : 1);                                         // chunks 0 .. nch-1 <-> calls call0 .. call0+nch-1
    const int base0 = (call0 + 1) * c.Nbits;                                         // bit position of chunk 0's first position
    const int w0 = base0 >> 5, nw = ((base0 + nch * c.Nbits + 31) >> 5) - w0 + 2;    // words covering them, + the funnel's second word
    uint32_t *s_key = s_w + (((kUwCalls + K + 1) * c.Nbits + 31) / 32 + 4);          // [nch][2]: whole chunk, head
    const uint32_t *src = words + (size_t)s * nwords;
    for (int i = threadIdx.x; i < nw; i += 256) s_w[i] = (w0 + i < nwords) ? src[w0 + i] : 0u;
    __syncthreads();
    const int sub = threadIdx.x & (kUwLanes - 1);
    for (int ch = threadIdx.x / kUwLanes; ch < nch; ch += 256 / kUwLanes) {
        const int pb = base0 + ch * c.Nbits;                                         // the chunk's first position
        uint32_t ka = 0xffffffffu, kh = 0xffffffffu;
        for (int i = sub; i < c.Nbits; i += kUwLanes) {
            const int p = pb + i;
            const uint32_t e = p + 32 <= nbits_total ? (uint32_t)__popc(window32(s_w, p - 32 * w0) ^ c.uw_word) : 255u;
            const uint32_t k = (e << 16) | (uint32_t)i;
            ka = k < ka ? k : ka;
            if (i < rem) kh = k < kh ? k : kh;
        }
        for (int o = kUwLanes / 2; o > 0; o >>= 1) {
            const uint32_t a = (uint32_t)__shfl_xor((int)ka, o, kWave), h = (uint32_t)__shfl_xor((int)kh, o, kWave);
            ka = a < ka ? a : ka; kh = h < kh ? h : kh;
        }
        if (sub == 0) { s_key[2 * ch] = ka; s_key[2 * ch + 1] = kh; }
    }
    __syncthreads();
    // (a chunk's key carries the position inside the chunk; chunk c+j sits j * Nbits into call c's window. Positions stay below
    //  bpf < 65536 and a chunk is never empty, so the add cannot carry into the error field.)
    for (int cl = threadIdx.x; cl < ncl; cl += 256) {
        uint32_t key = 0xffffffffu;
        for (int j = 0; j < K; j++) { const uint32_t k = s_key[2 * (cl + j)] + (uint32_t)(j * c.Nbits); key = k < key ? k : key; }
        if (rem) { const uint32_t k = s_key[2 * (cl + K) + 1] + (uint32_t)(K * c.Nbits); key = k < key ? k : key; }
        best[(size_t)s * ncalls + call0 + cl] = key;
    }
}

// ---- stage 2: sync state machine, one lane per stream ---------------------------------------------------------------------
// Window of call c (after its Nbits have been shifted in): stream bits [(c+1)*Nbits, (c+1)*Nbits + 2*bpf) of llr_all
// (the history occupies the first 2*bpf). [UPSTREAM-RECALLED codec2 freedv_fsk.c: freedv_rx_fsk_ldpc_data]
// Calls beyond ncalls_s[s] (the demodulator produced fewer frames for this stream than the batch is wide) are NOT demodulator
// calls: the state machine does not see them (status 0, info -1) and the history kept for the next batch ends at the last
// valid call -- upstream only ever advances its buffer on real demodulator output.
__global__ void fsm_kernel(LdpcDev c, int nstreams, int ncalls, const int32_t *ncalls_s, const uint32_t *words, int nwords, const uint32_t *best_key,
                           int nbits_total, FsmState *st, uint8_t *status, int32_t *info, int32_t *jobs, int32_t *njobs, int max_jobs)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstreams) return;
    FsmState f = st[s];
    const uint32_t *w = words + (size_t)s * nwords;
    int nj = 0;
    int valid = ncalls_s ? ncalls_s[s] : ncalls;
    valid = valid < 0 ? 0 : (valid > ncalls ? ncalls : valid);
    for (int call = valid; call < ncalls; call++) {
        status[(size_t)s * ncalls + call] = 0;
        int32_t *o = info + ((size_t)s * ncalls + call) * kInfoPerCall;
        for (int i = 0; i < kInfoPerCall; i++) o[i] = -1;
    }
    for (int call = 0; call < valid; call++) {
        const int base = (call + 1) * c.Nbits;             // stream-bit index of window position 0
        int next = f.state;
        if (f.state == 0) {
            const uint32_t key = best_key[(size_t)s * ncalls + call];
            const int best = (int)(key >> 16), bi = (int)(key & 0xffffu);
            f.uw_err = best;
            if (best <= c.uw_thresh1) { next = 1; f.loc = bi; f.bad_uw = 0; }
        } else {
            f.loc -= c.Nbits;
            if (f.loc < 0) {
                f.loc += c.bpf;
                f.uw_err = uw_errors(w, base + f.loc, nbits_total, c.uw_word);
                if (f.uw_err > c.uw_thresh2) { f.bad_uw++; if (f.bad_uw >= c.bad_uw_thresh) next = 0; }
                else f.bad_uw = 0;
            }
        }
        int stt = 0, pos = -1;
        if (next == 1) {
            stt |= kRxSync;
            if (f.loc >= 0 && f.loc < c.Nbits) {           // the frame is complete and about to slide out: decode it now
                pos = base + f.loc;
                if (nj < max_jobs) { jobs[((size_t)s * max_jobs + nj) * 2] = call; jobs[((size_t)s * max_jobs + nj) * 2 + 1] = pos; nj++; }
            }
        }
        f.state = next;
        status[(size_t)s * ncalls + call] = (uint8_t)stt;
        int32_t *o = info + ((size_t)s * ncalls + call) * kInfoPerCall;
        o[0] = f.state; o[1] = f.loc; o[2] = f.uw_err; o[3] = f.bad_uw; o[4] = 0; o[5] = 0; o[6] = pos >= 0 ? f.loc : -1; o[7] = 0; o[8] = 0; o[9] = 0;
    }
    st[s] = f;
    njobs[s] = nj;
}

// ---- stage 3: sum-product decode, one wave per frame ---------------------------------------------------------------------------
// dynamic LDS: [row_ptr m+1 | col_ptr n+1 | col_idx E | col_edge E] u16, [phi kPhiN] f32, per wave [Q n | r E] f32 + [hard n] u8
// (the channel LLRs are read from global memory where they are needed -- once per iteration and lane, L2-resident -- which
//  is what lets eight waves share one copy of H and two such workgroups share a CU)
template <int WPB, bool REGIDX>
__global__ __launch_bounds__(kWave * WPB, REGIDX ? 4 : 1) void decode_kernel(LdpcDev c, int njob_slots, const int32_t *jobs, const int32_t *njobs,
                                                             const h16 *llr_src, size_t llr_stride, int direct,
                                                             uint8_t *status, int ncalls, uint8_t *payload, int32_t *info,
                                                             uint8_t *cw_out, int32_t *iter_pcc_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *s_row_ptr = (uint16_t *)smem;
    uint16_t *s_col_ptr = s_row_ptr + (c.m + 1);
    uint16_t *s_col_idx = s_col_ptr + (c.n + 1);
    uint16_t *s_col_edge = s_col_idx + c.E;
    size_t off = (((size_t)(c.m + 1 + c.n + 1 + 2 * c.E) * 2) + 15) & ~(size_t)15;
    float *s_phi = (float *)(smem + off); off += (size_t)kPhiN * 4;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const size_t per_wave = ((size_t)(c.n + c.E) * 4 + (size_t)c.n + 15) & ~(size_t)15;
    float *Q = (float *)(smem + off + (size_t)wv * per_wave);
    float *r = Q + c.n;
    uint8_t *hard = (uint8_t *)(r + c.E);

    // which frames: direct mode = codeword indices (parity tests / library entry), else stream blockIdx.y's job list; a
    // workgroup walks its share of them WPB at a time, so H and the phi table are staged once per workgroup, not per frame
    const int s = blockIdx.y;
    const int nslots = direct ? njob_slots : njobs[s];
    if (blockIdx.x * WPB >= nslots) return;
    for (int i = threadIdx.x; i <= c.m; i += kWave * WPB) s_row_ptr[i] = c.row_ptr[i];
    for (int i = threadIdx.x; i <= c.n; i += kWave * WPB) s_col_ptr[i] = c.col_ptr[i];
    for (int i = threadIdx.x; i < c.E; i += kWave * WPB) { s_col_idx[i] = c.col_idx[i]; s_col_edge[i] = c.col_edge[i]; }
    for (int i = threadIdx.x; i < kPhiN; i += kWave * WPB) s_phi[i] = c.phi[i];
    __syncthreads();
    // REGIDX: this lane's rows (lane + 64 i) as registers
    int re0[REGIDX ? kRowsPerLane : 1], rdeg[REGIDX ? kRowsPerLane : 1];
    uint32_t rcol[REGIDX ? kRowsPerLane : 1][kDegFast / 2];
    if constexpr (REGIDX) {
#pragma unroll
        for (int i = 0; i < kRowsPerLane; i++) {
            const int row = lane + kWave * i;
            re0[i] = 0; rdeg[i] = 0;
            if (row < c.m) { re0[i] = s_row_ptr[row]; rdeg[i] = s_row_ptr[row + 1] - re0[i]; }
#pragma unroll
            for (int j = 0; j < kDegFast; j += 2) {
                const uint32_t lo = j < rdeg[i] ? s_col_idx[re0[i] + j] : 0u, hi = j + 1 < rdeg[i] ? s_col_idx[re0[i] + j + 1] : 0u;
                rcol[i][j / 2] = lo | (hi << 16);
            }
        }
    }

    for (int slot = blockIdx.x * WPB + wv; slot < nslots; slot += gridDim.x * WPB) {
    int call = 0;
    const h16 *src;
    if (direct) {
        src = llr_src + (size_t)slot * c.n;
    } else {
        call = jobs[((size_t)s * njob_slots + slot) * 2];
        const int pos = jobs[((size_t)s * njob_slots + slot) * 2 + 1];
        src = llr_src + (size_t)s * llr_stride + pos + kUwBits;      // codeword LLRs follow the unique word
    }
    const h16 *llr = src;
    for (int v = lane; v < c.n; v += kWave) Q[v] = h2f(llr[v]);
    for (int e = lane; e < c.E; e += kWave) r[e] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int iter = 0, pcc = 0;
    for (int it = 1; it <= c.max_iter; it++) {
        // check nodes: r_e = (product of the other signs) * phi(sum of the other phi(|q|)), q = Q - r (old)
        if constexpr (REGIDX) {
#pragma unroll
            for (int i = 0; i < kRowsPerLane; i++) {
                float S = 0.0f, a[kDegFast];
                unsigned sg = 0, negs = 0;
#pragma unroll
                for (int j = 0; j < kDegFast; j++) {
                    a[j] = 0.0f;
                    if (j < rdeg[i]) {
                        const int col = (int)((rcol[i][j / 2] >> (16 * (j & 1))) & 0xffffu);
                        const float q = Q[col] - r[re0[i] + j];
                        const unsigned ng = (q < 0.0f) ? 1u : 0u;
                        sg ^= ng; negs |= ng << j;
                        a[j] = phi_lookup(s_phi, fabsf(q));
                        S = S + a[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < kDegFast; j++)
                    if (j < rdeg[i]) {
                        const float mag = phi_lookup(s_phi, S - a[j]);
                        r[re0[i] + j] = (sg ^ ((negs >> j) & 1u)) ? -mag : mag;
                    }
                __builtin_amdgcn_sched_barrier(0);         // one row at a time: interleaving the unrolled rows costs 100 VGPRs
            }
        } else if (c.max_row_deg <= kDegFast) {
            // the same arithmetic with each edge's phi(|q|) and sign kept in registers between the two passes
            for (int row = lane; row < c.m; row += kWave) {
                const int e0 = s_row_ptr[row], e1 = s_row_ptr[row + 1];
                float S = 0.0f, a[kDegFast];
                unsigned sg = 0, negs = 0;
#pragma unroll
                for (int j = 0; j < kDegFast; j++) {
                    a[j] = 0.0f;
                    if (e0 + j < e1) {
                        const float q = Q[s_col_idx[e0 + j]] - r[e0 + j];
                        const unsigned ng = (q < 0.0f) ? 1u : 0u;
                        sg ^= ng; negs |= ng << j;
                        a[j] = phi_lookup(s_phi, fabsf(q));
                        S = S + a[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < kDegFast; j++)
                    if (e0 + j < e1) {
                        const float mag = phi_lookup(s_phi, S - a[j]);
                        r[e0 + j] = (sg ^ ((negs >> j) & 1u)) ? -mag : mag;
                    }
            }
        } else
        for (int row = lane; row < c.m; row += kWave) {
            const int e0 = s_row_ptr[row], e1 = s_row_ptr[row + 1];
            float S = 0.0f;
            unsigned sg = 0;
            for (int e = e0; e < e1; e++) {
                const float q = Q[s_col_idx[e]] - r[e];
                sg ^= (q < 0.0f) ? 1u : 0u;
                S = S + phi_lookup(s_phi, fabsf(q));
            }
            for (int e = e0; e < e1; e++) {
                const float q = Q[s_col_idx[e]] - r[e];
                const float a = phi_lookup(s_phi, fabsf(q));
                const float mag = phi_lookup(s_phi, S - a);
                const unsigned neg = sg ^ ((q < 0.0f) ? 1u : 0u);
                r[e] = neg ? -mag : mag;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // variable nodes: Q = llr + sum of incoming (ascending check order)
        for (int v = lane; v < c.n; v += kWave) {
            float acc = h2f(llr[v]);
            for (int j = s_col_ptr[v]; j < s_col_ptr[v + 1]; j++) acc = acc + r[s_col_edge[j]];
            Q[v] = acc;
            hard[v] = acc < 0.0f ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int ok = 0;
        if constexpr (REGIDX) {
#pragma unroll
            for (int i = 0; i < kRowsPerLane; i++) {
                unsigned x = 0;
#pragma unroll
                for (int j = 0; j < kDegFast; j++)
                    if (j < rdeg[i]) x ^= hard[(rcol[i][j / 2] >> (16 * (j & 1))) & 0xffffu];
                ok += (lane + kWave * i < c.m) && !x;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int row = lane; row < c.m; row += kWave) {
            unsigned x = 0;
            for (int e = s_row_ptr[row]; e < s_row_ptr[row + 1]; e++) x ^= hard[s_col_idx[e]];
            ok += !x;
        }
        for (int o = 32; o > 0; o >>= 1) ok += __shfl_xor(ok, o, kWave);
        iter = it; pcc = ok;
        if (ok == c.m) break;
    }

    // channel hard decisions that the decoder changed ("eraw" of rtl_fsk's -v line when the frame decodes)
    int eraw = 0;
    for (int v = lane; v < c.n; v += kWave) eraw += (int)((h2f(llr[v]) < 0.0f) != (hard[v] != 0));
    for (int o = 32; o > 0; o >>= 1) eraw += __shfl_xor(eraw, o, kWave);
    if (direct) {
        for (int v = lane; v < c.n; v += kWave) cw_out[(size_t)slot * c.n + v] = hard[v];
        if (lane == 0) { iter_pcc_out[2 * slot] = iter; iter_pcc_out[2 * slot + 1] = pcc; }
        continue;
    }
    // payload bytes (MSB first), CRC16 over all but the last two, status flags
    const int nbytes = c.k / 8;
    uint8_t *pl = payload + ((size_t)s * ncalls + call) * nbytes;
    for (int b = lane; b < nbytes; b += kWave) {
        unsigned byte = 0;
        for (int i = 0; i < 8; i++) byte |= (unsigned)hard[8 * b + i] << (7 - i);
        pl[b] = (uint8_t)byte;
        hard[c.n - nbytes + b] = (uint8_t)byte;                 // parity-bit area reused as a byte buffer for the CRC (n - k >= k/8)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        const uint8_t *bytes = hard + c.n - nbytes;
        uint16_t crc = 0xFFFF;
        for (int i = 0; i < nbytes - 2; i++) {
            uint8_t x = (uint8_t)(crc >> 8) ^ bytes[i];
            x ^= x >> 4;
            crc = (uint16_t)((crc << 8) ^ ((uint16_t)x << 12) ^ ((uint16_t)x << 5) ^ (uint16_t)x);
        }
        const bool crc_ok = crc == (uint16_t)((bytes[nbytes - 2] << 8) | bytes[nbytes - 1]);
        uint8_t stt = status[(size_t)s * ncalls + call];
        if (crc_ok) stt |= kRxBits;
        if (pcc != c.m) stt |= kRxBitErrors;
        status[(size_t)s * ncalls + call] = stt;
        int32_t *o = info + ((size_t)s * ncalls + call) * kInfoPerCall;
        o[4] = iter; o[5] = pcc; o[7] = crc_ok ? 1 : 0; o[8] = eraw;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }   // frames of this wave
}

// fused path: last batch's two frames of soft bits in front of this batch's, and their hard-decision words (2 bpf is a whole number of words)
__global__ void hist_prepare_kernel(int bpf, const h16 *llr_hist, h16 *llr_all, size_t llr_stride, uint32_t *words, int nwords)
{
    const int s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const h16 *hs = llr_hist + (size_t)s * 2 * bpf;
    if (i < 2 * bpf) llr_all[(size_t)s * llr_stride + i] = hs[i];
    if (i < (2 * bpf) / 32) {
        uint32_t v = 0;
        for (int b = 0; b < 32; b++) if (h2f(hs[32 * i + b]) < 0.0f) v |= 0x80000000u >> b;
        words[(size_t)s * nwords + i] = v;
    }
}

// ---- stage 3, codes that fit kFastRows x kFastVars with row weight <= 8 and column weight <= 4 (the FSK_LDPC code's shape) -----------
// The same flooding sum-product, operation for operation, in the storage layout of fsk_ldpc.hpp: DecoderLayout. The kernel is bound
// by VALU issue (PMC, profiles/r03_*: each wave executes a VALU instruction 19 % of its cycles, four waves per SIMD: 77 % of the
// pipe), so everything here is about instructions per edge:
//   * every index list lives in registers for the workgroup's life, ALREADY AS LDS BYTE ADDRESSES (packed u16): a lane's 4 check
//     rows (where each of their columns' Q lives), its 8 variables (where each incoming message lives, where the variable sits
//     in the codeword) -- an access is one unpack and one ds_read;
//   * NO per-edge predication: a row's unused slots point at a Q entry that holds +1e30 -- phi(|1e30 - anything|) is the table's
//     exact 0 for x >= 32, which adds nothing to the row's sum, and its sign bit is clear, which xors nothing into the row's
//     sign word; a variable's unused slots point at a message that stays +0 (x + 0 = x; a sum that is -0 is stored as +0, see
//     below). Their stores land in message slots no variable refers to. (Predicated, hipcc wraps every edge in an exec-mask
//     region and waits for its reads before the next.)
//   * messages are slot-major (slot j of the row at position p at j * 256 + p), Q and the binary16 channel LLRs are indexed by
//     storage position: every read and write of a wave is lane-consecutive except the two gathers, whose bank pattern the host
//     has spread (make_decoder_layout);
//   * signs ride in sign bits: "q < 0" is the sign bit of q = Q - r (never -0: Q is stored canonical, see below), a row's sign
//     product the xor of those words, an edge's own sign sits in the (otherwise clear) sign bit of its phi term, "-mag" is mag
//     with the sign bit set; Q is stored as sum + 0, so that its sign bit IS the hard decision "sum < 0" and the parity pass is
//     an xor of the words the check pass reads anyway;
//   * phi(x) is one float clamp to [2^-24, 32], a bit-field extract and a shift-add: the clamp's upper end lands on an extra
//     table entry that holds 0 -- the values phi_lookup returns.
// Bit for bit what the comparisons, negations and predicated loops of decode_kernel / the checker (ldpc_oracle.c) give (tested).
// LDS: per wave Q (2 KB), messages (MAXDEG KB), LLRs (1 KB), then the phi table and the slot counter: 4 waves = 40 KB at row weight 6,
// four workgroups per CU.
struct FastDev { const uint16_t *rcol, *vedge, *vsrc; int maxdeg; };
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ float lds_ld(uint32_t a) { return *(lds_f32 *)(uintptr_t)a; }
__device__ __forceinline__ void lds_st(uint32_t a, float v) { *(lds_f32 *)(uintptr_t)a = v; }

template <int WPB, int MAXDEG, int MAXCOL>
// (row weight 7-8: 11.3 KB of LDS per wave hold the CU at 12 waves whatever the registers -- the build for that shape may use 168 VGPRs (at 128 it spilled 18))
__global__ __launch_bounds__(kWave * WPB, MAXDEG > 6 ? 3 : 4) void decode_fast_kernel(LdpcDev c, FastDev fd, int njob_slots, const int32_t *jobs, const int32_t *njobs,
                                                                  const h16 *llr_src, size_t llr_stride, int direct,
                                                                  uint8_t *status, int ncalls, uint8_t *payload, int32_t *info,
                                                                  uint8_t *cw_out, int32_t *iter_pcc_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPL = kFastRows / kWave, VPL = kFastVars / kWave;             // 4 rows, 8 variables per lane
    constexpr int QN = kFastVars + 4, RN = MAXDEG * kFastRows + 4;              // + the neutral entries
    constexpr size_t per_wave = (size_t)QN * 4 + (size_t)RN * 4 + (size_t)kFastVars * 2;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // (the per-wave regions come first and the phi table behind them: its base minus the first bin's offset is then a non-negative
    //  compile-time constant for the shapes that matter and rides in the ds_read's offset field -- one add less per look-up)
    const uint32_t phi_a = lds0 + (uint32_t)WPB * (uint32_t)per_wave;           // [kPhiN + 4] floats, then the workgroup's slot counter
    const uint32_t q_a = lds0 + (uint32_t)wv * (uint32_t)per_wave;              // Q[QN]: [kFastVars] = +1e30 (neutral column)
    const uint32_t r_a = q_a + QN * 4;                                          // messages [MAXDEG][256]; [MAXDEG * 256] = +0 (neutral message)
    const uint32_t l_a = r_a + RN * 4;                                          // binary16 channel LLRs by storage index
    float *s_phi = (float *)(smem + (phi_a - lds0));
    int *s_next = (int *)(smem + (phi_a - lds0) + (size_t)(kPhiN + 4) * 4);
    float *Q = (float *)(smem + (q_a - lds0));
    float *r = (float *)(smem + (r_a - lds0));
    h16 *L16 = (h16 *)(smem + (l_a - lds0));
    uint8_t *hard = (uint8_t *)r;                                               // [n] by codeword position, after the iterations (messages are dead)

    const int s = blockIdx.y;
    const int nslots = direct ? njob_slots : njobs[s];
    if (blockIdx.x * WPB >= nslots) return;
    for (int i = threadIdx.x; i < kPhiN + 4; i += kWave * WPB) s_phi[i] = i < kPhiN ? c.phi[i] : 0.0f;     // (bins from x = 10 on hold 0: phi(x >= 10) = 0)
    if (threadIdx.x == 0) *s_next = 0;
    // this lane's rows (positions lane + 64 i) and variables (storage indices lane + 64 k): LDS byte addresses, two per register
    uint32_t rc[RPL][MAXDEG / 2], ve[VPL][(MAXCOL + 1) / 2], vs[VPL / 2];
    int rvalid = 0;                                                             // bit i: position lane + 64 i holds a row
#pragma unroll
    for (int i = 0; i < RPL; i++) {
        const uint16_t *src = fd.rcol + (size_t)(lane + kWave * i) * kFastRowDeg;
#pragma unroll
        for (int j = 0; j < MAXDEG; j += 2) {
            const uint32_t c0 = src[j], c1 = src[j + 1];
            rc[i][j / 2] = (q_a + 4u * (c0 != 0xffffu ? c0 : (uint32_t)kFastVars)) | ((q_a + 4u * (c1 != 0xffffu ? c1 : (uint32_t)kFastVars)) << 16);
        }
        rvalid |= (src[0] != 0xffffu) << i;
    }
#pragma unroll
    for (int k = 0; k < VPL; k++) {
        const uint16_t *src = fd.vedge + (size_t)(lane + kWave * k) * kFastColDeg;
#pragma unroll
        for (int t = 0; t < MAXCOL; t += 2) {
            const uint32_t e0 = src[t], e1 = t + 1 < MAXCOL ? src[t + 1] : 0xffffu;
            ve[k][t / 2] = (r_a + 4u * (e0 != 0xffffu ? e0 : (uint32_t)(MAXDEG * kFastRows))) | ((r_a + 4u * (e1 != 0xffffu ? e1 : (uint32_t)(MAXDEG * kFastRows))) << 16);
        }
    }
#pragma unroll
    for (int k = 0; k < VPL; k += 2) vs[k / 2] = (uint32_t)fd.vsrc[lane + kWave * k] | ((uint32_t)fd.vsrc[lane + kWave * (k + 1)] << 16);
    __syncthreads();
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // phi table look-up as an LDS address: clamp, exponent + 5 mantissa bits, x 4
    // (the clamp takes |x| as a source modifier; exponent + five mantissa bits are bits 18..30; base and first bin folded into one add)
    // (this kernel has no static LDS, so its dynamic LDS starts at address 0 -- checked below, the kernel traps otherwise -- and the
    //  table base is written as a literal: hipcc keeps the dynamic-LDS symbol opaque until link time and would add it per look-up)
    constexpr uint32_t kPhiFirst = 4u * ((uint32_t)(127 + kPhiLoExp) << 5);
    constexpr bool kPhiLit = (uint32_t)WPB * (uint32_t)per_wave >= kPhiFirst;
    // (pirip_hip_ldpc_create checks the assumption on the host -- hipFuncGetAttributes: this kernel has no static LDS -- and sends the handle
    //  to the generic decoder otherwise; the trap is the last line of defence, not the mechanism)
    if (kPhiLit && lds0 != 0) __builtin_trap();
    const uint32_t phi_b = kPhiLit ? (uint32_t)WPB * (uint32_t)per_wave - kPhiFirst : phi_a - kPhiFirst;
    auto phi_at = [&](float x) {
        x = __builtin_fminf(__builtin_fmaxf(__builtin_fabsf(x), kPhiXLo), kPhiXHi);
        return lds_ld((__builtin_amdgcn_ubfe(__builtin_bit_cast(uint32_t, x), 18, 13) << 2) + phi_b);
    };
    // The kernel is bound by the LDS pipe (PMC, profiles/r05_o_configs_pmc.txt: SQ_LDS_IDX_ACTIVE 90 % of the cycles, 41 % of them
    // bank conflicts -- the data-dependent phi look-ups: 32 lanes of a group on 32 banks, the bank is the argument's top five
    // mantissa bits), the vector-memory path is idle. The first PIRIP_PHI_VMEM slots of a row's first-stage look-ups read the SAME
    // table from global memory (2.3 KB, L1-resident): same values, LDS pipe relieved. Measured (profiles/r05_q_phi_vmem_ab.txt):
    // receive stage at 3.5 dB 9.64 ms -> 9.35 / 9.28 / 9.26 / 9.23 for 3 / 4 / 5 / 6 slots, second-stage look-ups as well 10.4 (2nd only)
    // and 13.1 ms (all 48: the texture path then is the bottleneck); at 7 dB (1.2 iterations per frame) 4.13 -> 4.19 ms.
#ifndef PIRIP_PHI_VMEM
#define PIRIP_PHI_VMEM 4
#endif
    const __amdgpu_buffer_rsrc_t phi_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)c.phi - kPhiFirst), 0, (int)(kPhiFirst + (uint32_t)kPhiN * 4u), 0x00020000);
    auto phi_vm = [&](float x) {
        x = __builtin_fminf(__builtin_fmaxf(__builtin_fabsf(x), kPhiXLo), kPhiXHi);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(phi_rsrc, (int)(__builtin_amdgcn_ubfe(__builtin_bit_cast(uint32_t, x), 18, 13) << 2), 0, 0));
    };

    // The workgroup's frames (slots blockIdx.x * WPB + i + k * gridDim.x * WPB) are handed to whichever wave is free: frames that do
    // not converge take max_iter iterations, ones that do a handful, and a fixed share per wave leaves waves idle behind the
    // unluckiest one while the workgroup holds its LDS. A frame's result does not depend on the wave that decodes it.
    for (;;) {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)(uintptr_t)(phi_a + (uint32_t)(kPhiN + 4) * 4), 1,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        const int slot = blockIdx.x * WPB + (t % WPB) + (t / WPB) * (int)gridDim.x * WPB;
        if (slot >= nslots) break;
        int call = 0;
        const h16 *llr;
        if (direct) llr = llr_src + (size_t)slot * c.n;
        else {
            call = jobs[((size_t)s * njob_slots + slot) * 2];
            llr = llr_src + (size_t)s * llr_stride + jobs[((size_t)s * njob_slots + slot) * 2 + 1] + kUwBits;    // codeword LLRs follow the unique word
        }
        // channel LLRs to their storage positions (one gather from L2 per frame), messages to zero, the neutral entries
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const uint32_t v = (vs[k / 2] >> (16 * (k & 1))) & 0xffffu;
            const h16 x = v != 0xffffu ? llr[v] : (h16)0;
            L16[lane + kWave * k] = x;
            Q[lane + kWave * k] = h2f(x) + 0.0f;
        }
#pragma unroll
        for (int j = 0; j < MAXDEG; j++)
#pragma unroll
            for (int i = 0; i < RPL; i++) r[j * kFastRows + lane + kWave * i] = 0.0f;
        if (lane < 4) { Q[kFastVars + lane] = 1e30f; r[MAXDEG * kFastRows + lane] = 0.0f; }
        wsync();

        int iter = 0, pcc = 0;
        for (int it = 1; it <= c.max_iter; it++) {
            // (the packed address registers are made opaque once per iteration: otherwise hipcc hoists all the unpacked addresses
            //  out of the loop as loop invariants and spills)
#pragma unroll
            for (int i = 0; i < RPL; i++)
#pragma unroll
                for (int x = 0; x < MAXDEG / 2; x++) asm volatile("" : "+v"(rc[i][x]));
#pragma unroll
            for (int k = 0; k < VPL; k++)
#pragma unroll
                for (int x = 0; x < (MAXCOL + 1) / 2; x++) asm volatile("" : "+v"(ve[k][x]));
            // check nodes: r_e = (product of the other signs) * phi(sum of the other phi(|q|)), q = Q - r (old): three stages per pair
            // of rows, each stage's LDS reads in flight together
            // (two rows at a time: all four at once need more registers than the budget of 128 holds)
#pragma unroll
            for (int g = 0; g < RPL; g += 2) {
                uint32_t a[2][MAXDEG];
                float S[2];
                uint32_t sg[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t ra = r_a + 4u * (uint32_t)(lane + kWave * (g + i));
#pragma unroll
                    for (int j = 0; j < MAXDEG; j++) {
                        const uint32_t qa = (j & 1) ? (rc[g + i][j / 2] >> 16) : (rc[g + i][j / 2] & 0xffffu);
                        // (never -0: Q is stored canonical and x - x is +0, so "q < 0" is q's sign bit as it stands)
                        a[i][j] = __builtin_bit_cast(uint32_t, lds_ld(qa) - lds_ld(ra + 4u * kFastRows * j));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    S[i] = 0.0f; sg[i] = 0;
#pragma unroll
                    for (int j = 0; j < MAXDEG; j++) {
                        const uint32_t qb = a[i][j];
                        const float ph = j < PIRIP_PHI_VMEM ? phi_vm(__builtin_bit_cast(float, qb)) : phi_at(__builtin_bit_cast(float, qb));
                        sg[i] ^= qb;
                        S[i] = S[i] + ph;
                        a[i][j] = __builtin_bit_cast(uint32_t, ph) | (qb & 0x80000000u);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t ra = r_a + 4u * (uint32_t)(lane + kWave * (g + i));
#pragma unroll
                    for (int j = 0; j < MAXDEG; j++) {
                        const float sx = S[i] - __builtin_bit_cast(float, a[i][j] & 0x7fffffffu);
                        const float mag = phi_at(sx);
                        lds_st(ra + 4u * kFastRows * j, __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, mag) | ((sg[i] ^ a[i][j]) & 0x80000000u)));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wsync();
            // variable nodes: Q = llr + sum of incoming (ascending check order), stored + 0: the sign bit of Q is then "sum < 0"
#pragma unroll
            for (int k = 0; k < VPL; k++) {
                float in[MAXCOL];
#pragma unroll
                for (int t = 0; t < MAXCOL; t++) in[t] = lds_ld((t & 1) ? (ve[k][t / 2] >> 16) : (ve[k][t / 2] & 0xffffu));
                float acc = h2f(L16[lane + kWave * k]);
#pragma unroll
                for (int t = 0; t < MAXCOL; t++) acc = acc + in[t];
                Q[lane + kWave * k] = acc + 0.0f;
            }
            wsync();
            // parity checks: xor of the sign bits of a row's columns (the neutral column's is clear)
            int ok = 0;
#pragma unroll
            for (int i = 0; i < RPL; i++) {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < MAXDEG; j++) x ^= __builtin_bit_cast(uint32_t, lds_ld((j & 1) ? (rc[i][j / 2] >> 16) : (rc[i][j / 2] & 0xffffu)));
                ok += ((rvalid >> i) & 1) & (int)(~x >> 31);
            }
            for (int o = 32; o > 0; o >>= 1) ok += __shfl_xor(ok, o, kWave);
            iter = it; pcc = ok;
            if (ok == c.m) break;
        }

        // channel hard decisions that the decoder changed, and the decoded word in codeword order (the message array is free now)
        int eraw = 0;
        wsync();
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const uint32_t v = (vs[k / 2] >> (16 * (k & 1))) & 0xffffu;
            const uint32_t bit = __builtin_bit_cast(uint32_t, Q[lane + kWave * k]) >> 31;
            if (v != 0xffffu) { eraw += (int)((h2f(L16[lane + kWave * k]) < 0.0f) != (bit != 0)); hard[v] = (uint8_t)bit; }
        }
        for (int o = 32; o > 0; o >>= 1) eraw += __shfl_xor(eraw, o, kWave);
        wsync();
        if (direct) {
            for (int v = lane; v < c.n; v += kWave) cw_out[(size_t)slot * c.n + v] = hard[v];
            if (lane == 0) { iter_pcc_out[2 * slot] = iter; iter_pcc_out[2 * slot + 1] = pcc; }
            wsync();
            continue;
        }
        // payload bytes (MSB first), CRC16 over all but the last two, status flags
        const int nbytes = c.k / 8;
        uint8_t *pl = payload + ((size_t)s * ncalls + call) * nbytes;
        uint8_t *pbytes = (uint8_t *)Q;                        // Q is dead too: the packed payload for the CRC
        unsigned mybyte[2] = {0, 0};
        for (int b = lane, x = 0; b < nbytes && x < 2; b += kWave, x++) {
            unsigned byte = 0;
            for (int i = 0; i < 8; i++) byte |= (unsigned)hard[8 * b + i] << (7 - i);
            mybyte[x] = byte;
        }
        wsync();
        for (int b = lane, x = 0; b < nbytes && x < 2; b += kWave, x++) { pl[b] = (uint8_t)mybyte[x]; pbytes[b] = (uint8_t)mybyte[x]; }
        wsync();
        if (lane == 0) {
            uint16_t crc = 0xFFFF;
            for (int i = 0; i < nbytes - 2; i++) {
                uint8_t x = (uint8_t)(crc >> 8) ^ pbytes[i];
                x ^= x >> 4;
                crc = (uint16_t)((crc << 8) ^ ((uint16_t)x << 12) ^ ((uint16_t)x << 5) ^ (uint16_t)x);
            }
            const bool crc_ok = crc == (uint16_t)((pbytes[nbytes - 2] << 8) | pbytes[nbytes - 1]);
            uint8_t stt = status[(size_t)s * ncalls + call];
            if (crc_ok) stt |= kRxBits;
            if (pcc != c.m) stt |= kRxBitErrors;
            status[(size_t)s * ncalls + call] = stt;
            int32_t *o = info + ((size_t)s * ncalls + call) * kInfoPerCall;
            o[4] = iter; o[5] = pcc; o[7] = crc_ok ? 1 : 0; o[8] = eraw;
        }
        wsync();
    }   // frames of this wave
}

// ---- stage 3 for batches that fill the chip: decode_bank_kernel ----------------------------------------------------------------
// decode_fast_kernel is bound by the LDS pipe, and 41 % of its LDS cycles are bank conflicts of the data-dependent phi look-ups
// (profiles/r05_o_configs_pmc.txt). Here the table is REPLICATED ONCE PER BANK: entry (bin, c) at dword bin * 32 + c, lane l reads copy
// c = l & 31 -- a ds_read_b32 serves lanes {0-31} and {32-63} as two groups and lane l of either group can only ever touch bank l & 31:
// no look-up conflicts, whatever the arguments (2 LDS cycles instead of ~9). 576 bins x 128 B = 72 KB, so ONE persistent 8-wave workgroup
// per CU owns the table and walks frames of all streams. With that the phi look-ups are a quarter of the LDS time instead of two
// thirds and the kernel runs into VALU issue next (380 instructions per frame-iteration at 2.5 waves per SIMD: measured, round 6),
// so the 256-register budget of two waves per SIMD is spent on instructions and LDS traffic alike:
//   * rows in pairs, variables in pairs: every float32 add / subtract is one v_pk_add_f32 for two (same IEEE operation per half);
//   * xors three at a time (v_bitop3_b32), the look-up index as v_med3 + v_bfe + v_lshl_add;
//   * a row's old messages stay in the registers of the lane that wrote them (24 fewer ds_reads);
//   * the parity check of iteration it is the xor of the signs of the Q values that iteration it + 1's check pass reads anyway
//     (24 fewer gathers and the unpack / reduce around them): the loop leaves after that read when the word checks;
//   * the channel LLRs stay in registers; row addresses stay unpacked;
//   * within a stage all LDS reads are issued before the first store (hipcc orders loads behind stores it cannot prove disjoint:
//     look-up, store, look-up, ... costs one LDS round trip each -- 19 per iteration before, 6 now).
// 285 VALU + 132 LDS instructions per frame-iteration (decode_fast_kernel: 472 + 186). What bounds it now is again the LDS pipe, at its
// conflict-free rate plus the two gathers' residual conflicts (fsk_ldpc.cpp: make_decoder_layout; profiles/r06_*_decode_pmc.txt).
// Same operations on the same operands in the same order as decode_fast_kernel: hard outputs, iteration counts, parity-check counts and
// records are bit-identical (tested against it and against the mirror oracle).
// LDS: [phi 576 x 32 f32 | counter 16 B | per wave: Q 516 f32, messages MAXDEG x 256 + 4 f32]; no static LDS is assumed at address 0.
// Work: unit q = (stream, chunk of kBankChunk job slots); workgroup b owns units b, b + G, ... and its waves draw (unit, slot) pairs
// from an LDS counter -- no global atomics, frames of any stream go to whichever wave is free.
constexpr int kBankChunk = 16;
// Per frame (not per iteration) the persistent decoder additionally
//   * draws the NEXT frame's job and issues the gather of its channel LLRs before it starts iterating on this one (a wave has one
//     or two neighbours on its SIMD to hide a global-memory round trip behind, not three);
//   * checks the CRC in parallel: CRC-16/CCITT-FALSE is affine in the message bits, and a message that ends in its own CRC has
//     remainder 0 -- crc(word) = crc0 ^ xor over the set bits v of R[v], R[v] = the CRC (init 0) of the word with only bit v set.
//     Each lane xors the R of its own eight variables' bits (table by storage index, in registers), one wave xor-reduction:
//     the same verdict as the serial byte loop of decode_fast_kernel, 40 instructions instead of 30 dependent LDS reads;
//   * ors its status bits into the status byte's word with one no-return atomic instead of load / or / store.
struct BankDev { const uint16_t *vcrc; uint32_t crc0, cps, cps_magic; };   // cps: chunks per stream; cps_magic = ceil(2^32 / cps): unit / cps = mulhi(unit, magic)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (by value: __builtin_bit_cast applied directly to a vector element expression `v.y` reads element 0 with this hipcc)
__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
// xor of N words, three at a time (v_bitop3_b32 with the table of a ^ b ^ c)
template <int N> __device__ __forceinline__ uint32_t xor_all(const uint32_t (&x)[N])
{
    uint32_t acc = x[0];
    int j = 1;
#pragma unroll
    for (; j + 1 < N; j += 2) acc = __builtin_amdgcn_bitop3_b32(acc, x[j], x[j + 1], 0x96);
    if (j < N) acc ^= x[j];
    return acc;
}
template <int WPB, int MAXDEG, int MAXCOL>
__global__ __launch_bounds__(kWave * WPB, 1) void decode_bank_kernel(LdpcDev c, FastDev fd, BankDev bd, int njob_slots, int nstreams, const int32_t *jobs, const int32_t *njobs,
                                                                  const h16 *llr_src, size_t llr_stride, int direct,
                                                                  uint8_t *status, int ncalls, uint8_t *payload, int32_t *info,
                                                                  uint8_t *cw_out, int32_t *iter_pcc_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPL = kFastRows / kWave, VPL = kFastVars / kWave;             // 4 rows, 8 variables per lane
    constexpr int RP = RPL / 2, VP = VPL / 2;                                   // ... handled as pairs: packed float32 arithmetic (v_pk_add_f32)
    constexpr int QN = kFastVars + 4, RN = MAXDEG * kFastRows + 4;              // + the neutral entries
    constexpr uint32_t kPhiBytes = (uint32_t)kPhiN * 32u * 4u;
    constexpr uint32_t per_wave = (uint32_t)QN * 4u + (uint32_t)RN * 4u;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t cnt_a = lds0 + kPhiBytes;
    const uint32_t q_a = cnt_a + 16u + (uint32_t)wv * per_wave;                 // Q[QN]: [kFastVars ..] = +1e30 (neutral column)
    const uint32_t r_a = q_a + QN * 4;                                          // messages [MAXDEG][256]; [MAXDEG * 256 ..] = +0 (neutral message)
    float *s_phi = (float *)smem;
    float *Q = (float *)(smem + (q_a - lds0));
    float *r = (float *)(smem + (r_a - lds0));
    uint8_t *hard = (uint8_t *)r;                                               // [n] by codeword position, after the iterations (messages are dead)

    for (int i = threadIdx.x; i < kPhiN * 32; i += kWave * WPB) s_phi[i] = c.phi[i >> 5];
    if (threadIdx.x == 0) *(int *)(smem + kPhiBytes) = 0;
    // this lane's rows (positions lane + 64 i): LDS byte addresses of their columns' Q; its variables (storage indices lane + 64 k):
    // offsets of the incoming messages in the wave's message array, two per register, the codeword positions and the CRC terms
    uint32_t rc[RPL][MAXDEG], ve[VPL][(MAXCOL + 1) / 2], vs[VPL / 2], vcrc[VPL / 2];
    int rvalid = 0;
#pragma unroll
    for (int i = 0; i < RPL; i++) {
        const uint16_t *src = fd.rcol + (size_t)(lane + kWave * i) * kFastRowDeg;
#pragma unroll
        for (int j = 0; j < MAXDEG; j++) {
            const uint32_t c0 = src[j];
            rc[i][j] = q_a + 4u * (c0 != 0xffffu ? c0 : (uint32_t)kFastVars);
        }
        rvalid |= (src[0] != 0xffffu) << i;
    }
#pragma unroll
    for (int k = 0; k < VPL; k++) {
        const uint16_t *src = fd.vedge + (size_t)(lane + kWave * k) * kFastColDeg;
#pragma unroll
        for (int t = 0; t < MAXCOL; t += 2) {
            const uint32_t e0 = src[t], e1 = t + 1 < MAXCOL ? src[t + 1] : 0xffffu;
            // (byte offsets from r_a: this kernel's LDS addresses do not fit 16 bits)
            ve[k][t / 2] = (4u * (e0 != 0xffffu ? e0 : (uint32_t)(MAXDEG * kFastRows))) | ((4u * (e1 != 0xffffu ? e1 : (uint32_t)(MAXDEG * kFastRows))) << 16);
        }
    }
#pragma unroll
    for (int k = 0; k < VPL; k += 2) {
        vs[k / 2] = (uint32_t)fd.vsrc[lane + kWave * k] | ((uint32_t)fd.vsrc[lane + kWave * (k + 1)] << 16);
        vcrc[k / 2] = (uint32_t)bd.vcrc[lane + kWave * k] | ((uint32_t)bd.vcrc[lane + kWave * (k + 1)] << 16);
    }
    __syncthreads();
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // look-up address: clamp (one median), exponent + five mantissa bits = bits 18..30, x 128 B, + this lane's bank column -- the
    // table base and the first bin's offset are folded into the per-lane constant (32-bit wrap-around arithmetic)
    const uint32_t lane_b = lds0 + 4u * (uint32_t)(lane & 31) - (((uint32_t)(127 + kPhiLoExp) << 5) << 7);
    auto phi_at = [&](float x) {
        x = __builtin_amdgcn_fmed3f(__builtin_fabsf(x), kPhiXLo, kPhiXHi);
        uint32_t bin;       // (written as the instruction: hipcc otherwise folds the two shifts into shift + mask and needs a third operation for the add)
        asm("v_bfe_u32 %0, %1, 18, 13" : "=v"(bin) : "v"(x));
        return lds_ld((bin << 7) + lane_b);
    };
    // the next frame of this wave: stream, slot, demodulator call, LLRs; false when the workgroup's share is used up
    struct Job { int s, slot, call; const h16 *llr; };
    auto next_job = [&](Job &j) -> bool {
        for (;;) {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)(uintptr_t)cnt_a, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(t);
            const uint32_t unit = blockIdx.x + (uint32_t)(t / kBankChunk) * gridDim.x;      // (the host keeps units * cps below 2^32: the magic division is exact)
            j.s = bd.cps == 1 ? (int)unit : (int)__umulhi(unit, bd.cps_magic);
            if (j.s >= nstreams) return false;
            j.slot = (int)(unit - (uint32_t)j.s * bd.cps) * kBankChunk + (t % kBankChunk);
            if (j.slot >= (direct ? njob_slots : njobs[j.s])) continue;
            j.call = 0;
            if (direct) j.llr = llr_src + (size_t)j.slot * c.n;
            else {
                j.call = jobs[((size_t)j.s * njob_slots + j.slot) * 2];
                j.llr = llr_src + (size_t)j.s * llr_stride + jobs[((size_t)j.s * njob_slots + j.slot) * 2 + 1] + kUwBits;    // codeword LLRs follow the unique word
            }
            return true;
        }
    };
    Job cur, nxt;
    h16 nx[VPL];                                                                // the coming frame's channel LLRs by storage index, in flight
    auto gather = [&](const Job &j) {                                          // (unconditional loads: an absent variable reads position 0 and is zeroed at use)
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const uint32_t v = (vs[k / 2] >> (16 * (k & 1))) & 0xffffu;
            nx[k] = j.llr[v != 0xffffu ? v : 0u];
        }
    };
    bool have = next_job(nxt);
    if (have) gather(nxt);

    while (have) {
        cur = nxt;
        // channel LLRs to registers and, as the first Q, to their storage positions; the neutral entries
        f32x2 Lf[VP];
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const uint32_t v = (vs[k / 2] >> (16 * (k & 1))) & 0xffffu;
            const float x = v != 0xffffu ? h2f(nx[k]) : 0.0f;
            Lf[k / 2][k & 1] = x;
            Q[lane + kWave * k] = x + 0.0f;
        }
        if (lane < 4) { Q[kFastVars + lane] = 1e30f; r[MAXDEG * kFastRows + lane] = 0.0f; }
        have = next_job(nxt);
        if (have) gather(nxt);
        f32x2 rold[RP][MAXDEG];                                                 // this lane's rows' messages as last written (first: +0)
#pragma unroll
        for (int p2 = 0; p2 < RP; p2++)
#pragma unroll
            for (int j = 0; j < MAXDEG; j++) rold[p2][j] = f32x2{0.0f, 0.0f};
        wsync();

        int iter = 0, pcc = 0;
        for (int it = 1; c.max_iter > 0; it++) {
            // check nodes, stage 1: q = Q - r (old) for every edge of this lane's rows, all gathers in flight together; the signs of the Q
            // values are the hard decisions of the iteration before (Q is stored canonical: never -0). Rows in pairs (2p, 2p + 1).
            f32x2 q[RP][MAXDEG];
            uint32_t par[RPL];
#pragma unroll
            for (int p2 = 0; p2 < RP; p2++) {
                uint32_t qb0[MAXDEG], qb1[MAXDEG];
#pragma unroll
                for (int j = 0; j < MAXDEG; j++) {
                    const f32x2 qv = {lds_ld(rc[2 * p2][j]), lds_ld(rc[2 * p2 + 1][j])};
                    qb0[j] = fbits(qv.x); qb1[j] = fbits(qv.y);
                    q[p2][j] = qv - rold[p2][j];
                }
                par[2 * p2] = xor_all(qb0); par[2 * p2 + 1] = xor_all(qb1);
            }
            if (it > 1) {
                int ok = 0;
#pragma unroll
                for (int i = 0; i < RPL; i++) ok += __builtin_popcountll(__builtin_amdgcn_ballot_w64((((rvalid >> i) & 1) & (int)(~par[i] >> 31)) != 0));
                iter = it - 1; pcc = ok;
                if (ok == c.m || it > c.max_iter) break;
            }
            // stage 2: phi of every |q|, the row's sum (ascending slot order) and sign product; stage 3: r_e = (product of the other
            // signs) * phi(sum of the others' phi). Each stage for ALL of the lane's rows at once: every look-up of a stage in flight
            // together, and all of them before the first store (hipcc keeps LDS loads behind earlier LDS stores -- it cannot see that
            // the table and the messages do not overlap -- and a look-up, store, look-up, ... sequence is one LDS round trip each).
            f32x2 ph[RP][MAXDEG], S[RP];
            uint32_t sg[RPL];
#pragma unroll
            for (int p2 = 0; p2 < RP; p2++) {
                uint32_t qb0[MAXDEG], qb1[MAXDEG];
                S[p2] = f32x2{0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < MAXDEG; j++) {
                    qb0[j] = fbits(q[p2][j].x); qb1[j] = fbits(q[p2][j].y);
                    ph[p2][j] = f32x2{phi_at(q[p2][j].x), phi_at(q[p2][j].y)};
                    S[p2] = S[p2] + ph[p2][j];
                }
                sg[2 * p2] = xor_all(qb0); sg[2 * p2 + 1] = xor_all(qb1);
            }
            uint32_t m0[RP][MAXDEG], m1[RP][MAXDEG];
#pragma unroll
            for (int p2 = 0; p2 < RP; p2++)
#pragma unroll
                for (int j = 0; j < MAXDEG; j++) {
                    const f32x2 sx = S[p2] - ph[p2][j];
                    m0[p2][j] = fbits(phi_at(sx.x)); m1[p2][j] = fbits(phi_at(sx.y));
                }
#pragma unroll
            for (int p2 = 0; p2 < RP; p2++) {
                const uint32_t ra = r_a + 4u * (uint32_t)(lane + kWave * 2 * p2);
#pragma unroll
                for (int j = 0; j < MAXDEG; j++) {
                    const uint32_t r0 = (m0[p2][j] & 0x7fffffffu) | ((sg[2 * p2] ^ fbits(q[p2][j].x)) & 0x80000000u);
                    const uint32_t r1 = (m1[p2][j] & 0x7fffffffu) | ((sg[2 * p2 + 1] ^ fbits(q[p2][j].y)) & 0x80000000u);
                    rold[p2][j] = f32x2{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
                    lds_st(ra + 4u * kFastRows * j, __builtin_bit_cast(float, r0));
                    lds_st(ra + 4u * kFastRows * j + 4u * kWave, __builtin_bit_cast(float, r1));
                }
            }
            wsync();
            // variable nodes: Q = llr + sum of incoming (ascending check order), stored + 0: the sign bit of Q is then "sum < 0"
            // (again every gather before the first store)
            f32x2 in[VP][MAXCOL];
#pragma unroll
            for (int k2 = 0; k2 < VP; k2++)
#pragma unroll
                for (int t2 = 0; t2 < MAXCOL; t2++)
                    in[k2][t2] = f32x2{lds_ld(r_a + ((t2 & 1) ? (ve[2 * k2][t2 / 2] >> 16) : (ve[2 * k2][t2 / 2] & 0xffffu))),
                                       lds_ld(r_a + ((t2 & 1) ? (ve[2 * k2 + 1][t2 / 2] >> 16) : (ve[2 * k2 + 1][t2 / 2] & 0xffffu)))};
#pragma unroll
            for (int k2 = 0; k2 < VP; k2++) {
                f32x2 acc = Lf[k2];
#pragma unroll
                for (int t2 = 0; t2 < MAXCOL; t2++) acc = acc + in[k2][t2];
                acc = acc + f32x2{0.0f, 0.0f};
                Q[lane + kWave * 2 * k2] = acc.x; Q[lane + kWave * (2 * k2 + 1)] = acc.y;
            }
            wsync();
        }

        // channel hard decisions that the decoder changed, the decoded word in codeword order (the message array is free now) and the
        // word's CRC remainder from this lane's bits
        int eraw = 0;
        uint32_t crc_e = 0, crc_o = 0;
        wsync();
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const uint32_t v = (vs[k / 2] >> (16 * (k & 1))) & 0xffffu;
            const uint32_t qb = __builtin_bit_cast(uint32_t, Q[lane + kWave * k]);
            const uint32_t bit = qb >> 31;
            if (k & 1) crc_o ^= vcrc[k / 2] & (uint32_t)((int32_t)qb >> 31); else crc_e ^= vcrc[k / 2] & (uint32_t)((int32_t)qb >> 31);
            if (v != 0xffffu) { eraw += (int)((Lf[k / 2][k & 1] < 0.0f) != (bit != 0)); hard[v] = (uint8_t)bit; }
        }
        uint32_t crc = (crc_e & 0xffffu) ^ (crc_o >> 16);
        for (int o = 32; o > 0; o >>= 1) { eraw += __shfl_xor(eraw, o, kWave); crc ^= (uint32_t)__shfl_xor((int)crc, o, kWave); }
        crc ^= bd.crc0;
        wsync();
        if (direct) {
            for (int v = lane; v < c.n; v += kWave) cw_out[(size_t)cur.slot * c.n + v] = hard[v];
            if (lane == 0) { iter_pcc_out[2 * cur.slot] = iter; iter_pcc_out[2 * cur.slot + 1] = pcc; }
            wsync();
            continue;
        }
        // payload bytes (MSB first) and the status flags
        const int nbytes = c.k / 8;
        const size_t rec = (size_t)cur.s * ncalls + cur.call;
        uint8_t *pl = payload + rec * nbytes;
        for (int b = lane; b < nbytes; b += kWave) {
            unsigned byte = 0;
            for (int i = 0; i < 8; i++) byte |= (unsigned)hard[8 * b + i] << (7 - i);
            pl[b] = (uint8_t)byte;
        }
        if (lane == 0) {
            const bool crc_ok = crc == 0;
            const uint32_t stt = (crc_ok ? (uint32_t)kRxBits : 0u) | (pcc != c.m ? (uint32_t)kRxBitErrors : 0u);
            const uintptr_t sa = (uintptr_t)(status + rec);
            if (stt) (void)__hip_atomic_fetch_or((uint32_t *)(sa & ~(uintptr_t)3), stt << (8 * (sa & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int32_t *o = info + rec * kInfoPerCall;
            o[4] = iter; o[5] = pcc; o[7] = crc_ok ? 1 : 0; o[8] = eraw;
        }
        wsync();
    }   // frames of this wave
}

// stand-alone decode entry: caller's float LLRs into the decoder's input format
__global__ void f32_to_h16_kernel(const float *src, h16 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2h(src[i]);
}

__global__ void save_hist_kernel(const h16 *llr_all, size_t llr_stride, int ncalls, const int32_t *ncalls_s, int Nbits, int bpf, h16 *llr_hist)
{
    const int s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    int valid = ncalls_s ? ncalls_s[s] : ncalls;
    valid = valid < 0 ? 0 : (valid > ncalls ? ncalls : valid);
    if (i < 2 * bpf) llr_hist[(size_t)s * 2 * bpf + i] = llr_all[(size_t)s * llr_stride + (size_t)valid * Nbits + i];
}

}  // namespace

enum { kDecAuto = 0, kDecGeneric = 1, kDecFast = 2, kDecBank = 3 };
struct pirip_hip_ldpc {
    LdpcCode code;
    LdpcDev dev{};
    DecoderLayout layout;                      // fast decoder's storage layout (host), device copies below
    uint16_t *d_rcol = nullptr, *d_vedge = nullptr, *d_vsrc = nullptr;
    uint16_t *d_vcrc = nullptr; uint32_t crc0 = 0;   // persistent decoder: CRC term of the bit at each storage index, CRC of the all-zero word
    // two builds of the fast decoder: row weight <= 6 (the FSK_LDPC code's shape: 4 data ones + the accumulator's 2), or the limit 8
    int fast_deg() const { return layout.maxdeg <= 6 ? 6 : kFastRowDeg; }
    size_t fast_lds_bytes(int wpb) const
    {
        return (size_t)(kPhiN + 4) * 4 + 16 + (size_t)wpb * ((size_t)(kFastVars + 4) * 4 + (size_t)(fast_deg() * kFastRows + 4) * 4 + (size_t)kFastVars * 2);
    }
    int nstreams = 0, device = 0, last_hip = 0;
    // internal HIP streams and events of the fork / join paths (this handle's own: two receivers driven from two host threads do not
    // meet on them; made on first use, destroyed with the handle). Slots 0 / 1: the two stream ranges of one call (low / high
    // priority); slots 2 ..: the groups of pirip_hip_fsk_ldpc_rx_batch_groups (2: the last group, low priority; the others high) --
    // a group that splits again inside does so on its own handle's slots 0 / 1 and its own fork event.
    static constexpr int kSideSlots = 13, kGroupSlot0 = 2, kMidSlot = 12;       // (slot 12: the middle priority -- the last range of a call whose earlier ranges decode on slot 0)
    hipStream_t side[kSideSlots] = {};
    hipEvent_t ev_fork = nullptr, ev_gfork = nullptr, ev_join[kSideSlots] = {}, ev_mid[3] = {};
    int split_bounds[3] = {5, 0, 0}, n_bounds = 1;   // ... where the ranges end, in eighths of the streams (PIRIP_CHAIN_SPLIT_EIGHTHS="5" | "4,6" | "3,5,7" at create: experiments)
    int split_min = 4096;                      // streams from which pirip_hip_fsk_ldpc_rx_batch runs two ranges side by side (PIRIP_CHAIN_SPLIT_MIN at create; 0: never)
    int overlap_decoder_fast = -1;             // -1: decided per call (below). PIRIP_CHAIN_OVERLAP_DECODER=off | fast | fast-low at create: the ranges that decode beside a demodulator use decode_fast_kernel
                                               // (1), and do so on the lowest-priority stream while the last range runs at the middle priority (2)
    int in_group = 0;                          // set by pirip_hip_fsk_ldpc_rx_batch_groups around its inner calls: other groups' demodulators share the chip, the rule below does not hold
    int test_fail_range = -1;                  // PIRIP_CHAIN_TEST_FAIL=<0|1> at create: that range of a split call reports an error after the fork (tests of the join)
    int num_cu = 256;                          // compute units of the device (the persistent decoder launches one workgroup per CU)
    int fast_static_lds = 0;                   // static LDS bytes of the fast decoder's instantiations (must be 0: its table base is a literal); else the generic decoder serves
    int decoder_pref = 0;                      // kDecAuto, or what PIRIP_LDPC_DECODER / PIRIP_LDPC_GENERIC asked for at create
    uint16_t *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_col_edge = nullptr;
    float *d_lnI0 = nullptr, *d_phi = nullptr; uint16_t *d_llr_hist = nullptr;
    FsmState *d_fsm = nullptr;
    // per-batch work buffers (grown on demand)
    uint16_t *d_llr_all = nullptr; uint32_t *d_words = nullptr, *d_best = nullptr; int32_t *d_jobs = nullptr, *d_njobs = nullptr;
    size_t cap_calls = 0;
    float *d_filt_work = nullptr; size_t filt_cap = 0;   // pirip_hip_fsk_ldpc_rx_batch's magnitudes when the fused hand-over does not apply
    int last_path_fused = 0;
    // host staging for the one-stream convenience entry
    float *d_h_filt = nullptr; uint8_t *d_h_status = nullptr, *d_h_payload = nullptr; int32_t *d_h_info = nullptr; size_t h_cap = 0;
    // direct-decode staging
    uint16_t *d_dd_llr = nullptr; uint8_t *d_dd_bits = nullptr; int32_t *d_dd_ip = nullptr; size_t dd_cap = 0;
    size_t lds_bytes(int wpb) const
    {
        size_t off = (((size_t)(code.m + 1 + code.n + 1 + 2 * (int)code.col_idx.size()) * 2) + 15) & ~(size_t)15;
        off += (size_t)kPhiN * 4;
        const size_t per_wave = ((size_t)(code.n + (int)code.col_idx.size()) * 4 + (size_t)code.n + 15) & ~(size_t)15;
        return off + (size_t)wpb * per_wave;
    }
};

#define LCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { h->last_hip = (int)e_; return PIRIP_ERR_HIP; } } while (0)

namespace {

bool bind_dev(const pirip_hip_ldpc *h)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == h->device) return true;
    return hipSetDevice(h->device) == hipSuccess;
}

template <typename T>
bool up(T **dst, const void *src, size_t bytes)
{
    if (hipMalloc((void **)dst, bytes ? bytes : 16) != hipSuccess) return false;
    return !bytes || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
}

int launch_decode(pirip_hip_ldpc *h, int slots, int nstreams_y, const int32_t *jobs, const int32_t *njobs, const uint16_t *llr, size_t llr_stride,
                  int direct, uint8_t *status, int ncalls, uint8_t *payload, int32_t *info, uint8_t *cw, int32_t *ip, hipStream_t st, bool beside_demod = false)
{
    if (slots <= 0) return PIRIP_OK;
    // batches that give every wave of the chip several frames: the persistent decoder with the bank-private phi table.
    // beside_demod: this decode runs next to another stream range's demodulator (split chain). The persistent decoder takes whole CUs
    // (152 KB of LDS, 8 waves x 199 VGPR: it starts on a CU only when all three demodulator workgroups there have ended), the small one
    // (40 KB, 4 waves x <= 128 VGPR) fits beside two demodulator workgroups and its LDS-bound waves share their SIMDs with VALU-bound ones.
    const bool small_beside = beside_demod && h->decoder_pref == kDecAuto && h->fast_static_lds == 0;
    if (h->layout.ok && !small_beside && (h->decoder_pref == kDecBank || (h->decoder_pref == kDecAuto && (int64_t)slots * nstreams_y >= (int64_t)h->num_cu * 8 * 4))) {
        const int deg = h->fast_deg(), wpb = 8;
        const size_t lds = (size_t)kPhiN * 32 * 4 + 16 + (size_t)wpb * ((size_t)(kFastVars + 4) * 4 + (size_t)(deg * kFastRows + 4) * 4);
        const int cps = (slots + kBankChunk - 1) / kBankChunk;
        const int64_t units = (int64_t)cps * nstreams_y;
        const int64_t gx = std::min<int64_t>(units, h->num_cu);
        if ((units + gx) * cps >= ((int64_t)1 << 32)) return PIRIP_ERR_UNSUPPORTED;          // (the kernel's magic division; 2^32 / cps job chunks: never in practice)
        const dim3 g((unsigned)gx), b(kWave * wpb);
        const FastDev fd{h->d_rcol, h->d_vedge, h->d_vsrc, h->layout.maxdeg};
        const BankDev bk{h->d_vcrc, h->crc0, (uint32_t)cps, (uint32_t)((((uint64_t)1 << 32) + (uint64_t)cps - 1) / (uint64_t)cps)};
#define PIRIP_BANK_LAUNCH(W, D) do { \
        LCHK(hipFuncSetAttribute((const void *)decode_bank_kernel<W, D, kFastColDeg>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((decode_bank_kernel<W, D, kFastColDeg>), g, b, lds, st, h->dev, fd, bk, slots, nstreams_y, jobs, njobs, llr, llr_stride, direct, status, ncalls, payload, info, cw, ip); } while (0)
        if (deg == 6) PIRIP_BANK_LAUNCH(8, 6); else PIRIP_BANK_LAUNCH(8, kFastRowDeg);
#undef PIRIP_BANK_LAUNCH
        LCHK(hipGetLastError());
        return PIRIP_OK;
    }
    if (h->layout.ok && h->decoder_pref != kDecGeneric && h->fast_static_lds == 0) {
        // four waves per workgroup, four workgroups per CU (measured against 8 x 2, 6 x 2 at three waves per SIMD and 4 x 2 at two:
        // 16.1 / 16.4 / 21.5 / 16.1 ms for the receive stage at 3.5 dB, 6.1 / 6.8 / 7.6 / 6.1 ms at 7 dB -- profiles/r03_experiments.txt)
        int wpb = 4;
        while (wpb > 1 && h->fast_lds_bytes(wpb) > 40 * 1024) wpb >>= 1;
        const size_t lds = h->fast_lds_bytes(wpb);
        int gx = (slots + wpb - 1) / wpb;
        const int want = 8192 / (nstreams_y > 0 ? nstreams_y : 1);
        if (gx > want) gx = want < 1 ? 1 : want;
        const dim3 g(gx, nstreams_y), b(kWave * wpb);
        const FastDev fd{h->d_rcol, h->d_vedge, h->d_vsrc, h->layout.maxdeg};
#define PIRIP_FAST_LAUNCH2(W, D, C) do { \
        if (lds > 48 * 1024) LCHK(hipFuncSetAttribute((const void *)decode_fast_kernel<W, D, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((decode_fast_kernel<W, D, C>), g, b, lds, st, h->dev, fd, slots, jobs, njobs, llr, llr_stride, direct, status, ncalls, payload, info, cw, ip); } while (0)
#define PIRIP_FAST_LAUNCH(W) do { if (h->fast_deg() == 6) PIRIP_FAST_LAUNCH2(W, 6, kFastColDeg); else PIRIP_FAST_LAUNCH2(W, kFastRowDeg, kFastColDeg); } while (0)
        if (wpb == 4) PIRIP_FAST_LAUNCH(4); else if (wpb == 2) PIRIP_FAST_LAUNCH(2); else PIRIP_FAST_LAUNCH(1);
#undef PIRIP_FAST_LAUNCH
#undef PIRIP_FAST_LAUNCH2
        LCHK(hipGetLastError());
        return PIRIP_OK;
    }
    int wpb = 8;
    while (wpb > 1 && h->lds_bytes(wpb) > 80 * 1024) wpb >>= 1;        // two workgroups per CU where the code allows it
    while (wpb > 1 && h->lds_bytes(wpb) > 160 * 1024) wpb >>= 1;
    const size_t lds = h->lds_bytes(wpb);
    if (lds > 160 * 1024) return PIRIP_ERR_UNSUPPORTED;
    // enough workgroups to fill the chip several times over, each walking its stream's frames (tables staged once per workgroup)
    int gx = (slots + wpb - 1) / wpb;
    const int want = 8192 / (nstreams_y > 0 ? nstreams_y : 1);
    if (gx > want) gx = want < 1 ? 1 : want;
    const dim3 g(gx, nstreams_y), b(kWave * wpb);
    const LdpcDev &c = h->dev;
    const bool regidx = c.m <= kWave * kRowsPerLane && c.max_row_deg <= kDegFast;
#define PIRIP_DEC_LAUNCH2(W, R) do { \
        if (lds > 48 * 1024) LCHK(hipFuncSetAttribute((const void *)decode_kernel<W, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((decode_kernel<W, R>), g, b, lds, st, h->dev, slots, jobs, njobs, llr, llr_stride, direct, status, ncalls, payload, info, cw, ip); } while (0)
#define PIRIP_DEC_LAUNCH(W) do { if (regidx) PIRIP_DEC_LAUNCH2(W, true); else PIRIP_DEC_LAUNCH2(W, false); } while (0)
    if (wpb == 8) PIRIP_DEC_LAUNCH(8); else if (wpb == 4) PIRIP_DEC_LAUNCH(4); else if (wpb == 2) PIRIP_DEC_LAUNCH(2); else PIRIP_DEC_LAUNCH(1);
#undef PIRIP_DEC_LAUNCH2
#undef PIRIP_DEC_LAUNCH
    LCHK(hipGetLastError());
    return PIRIP_OK;
}

}  // namespace

extern "C" {

int pirip_hip_ldpc_create(const char *code_path, int M, int Nsym, int nstreams, int device, pirip_hip_ldpc **out)
{
    if (!code_path || !out || nstreams <= 0 || (M != 2 && M != 4) || Nsym <= 0) return PIRIP_ERR_BAD_ARG;
    *out = nullptr;
    pirip_hip_ldpc *h = new (std::nothrow) pirip_hip_ldpc();
    if (!h) return PIRIP_ERR_NOMEM;
    const std::string err = h->code.load(code_path);
    if (!err.empty()) { fprintf(stderr, "pirip_hip_ldpc_create: %s: %s\n", code_path, err.c_str()); delete h; return PIRIP_ERR_BAD_CONFIG; }
    const LdpcCode &c = h->code;
    const int Nbits = Nsym * (M == 2 ? 1 : 2);
    if (Nbits > c.bits_per_frame()) { delete h; return PIRIP_ERR_BAD_CONFIG; }   // the sync logic assumes < one frame of bits per call
    if (c.col_idx.size() > 65535 || h->lds_bytes(1) > 160 * 1024) { delete h; return PIRIP_ERR_UNSUPPORTED; }
    if (pirip_hip_device_count() <= 0) { delete h; return PIRIP_ERR_NO_DEVICE; }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { delete h; return PIRIP_ERR_NO_DEVICE; }
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return PIRIP_ERR_NO_DEVICE; }
    h->nstreams = nstreams;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->num_cu = cus;
        // which decoder kernel serves this handle (all three give the same records; the choice is read once, here)
        const char *pref = getenv("PIRIP_LDPC_DECODER");
        if (getenv("PIRIP_LDPC_GENERIC")) h->decoder_pref = kDecGeneric;
        else if (pref && !strcmp(pref, "generic")) h->decoder_pref = kDecGeneric;
        else if (pref && !strcmp(pref, "fast")) h->decoder_pref = kDecFast;
        else if (pref && !strcmp(pref, "bank")) h->decoder_pref = kDecBank;
        if (const char *e = getenv("PIRIP_CHAIN_SPLIT_MIN")) h->split_min = atoi(e);
        if (const char *e = getenv("PIRIP_CHAIN_TEST_FAIL")) h->test_fail_range = atoi(e);
        if (const char *e = getenv("PIRIP_CHAIN_OVERLAP_DECODER")) h->overlap_decoder_fast = !strcmp(e, "fast") ? 1 : !strcmp(e, "fast-low") ? 2 : 0;
        if (getenv("PIRIP_CHAIN_SPLIT_EIGHTHS") && h->overlap_decoder_fast < 0) h->overlap_decoder_fast = 0;      // (an explicit split is an experiment: no automatic choice on top of it)
        if (const char *e = getenv("PIRIP_CHAIN_SPLIT_EIGHTHS")) {
            int b[3] = {0, 0, 0};
            const int nb = sscanf(e, "%d,%d,%d", &b[0], &b[1], &b[2]);
            bool good = nb >= 1;
            for (int i = 0; i < nb; i++) good = good && b[i] >= 1 && b[i] <= 7 && (i == 0 || b[i] > b[i - 1]);
            if (good) { h->n_bounds = nb; for (int i = 0; i < nb; i++) h->split_bounds[i] = b[i]; }
        }
    }
    auto to16 = [](const std::vector<int32_t> &v) { return std::vector<uint16_t>(v.begin(), v.end()); };
    const auto rp = to16(c.row_ptr), ci = to16(c.col_idx), cp = to16(c.col_ptr), ce = to16(c.col_edge);
    // tables: double libm on the host, rounded to float (the oracle builds the same numbers the same way)
    std::vector<float> lnI0(kLnI0N + 2), phi(kPhiN);
    for (int j = 0; j <= kLnI0N + 1; j++) {
        const double x = j / 8.0;
        // ln I0 by its power series (x <= 32: terms stay far below overflow in double)
        double term = 1.0, sum = 1.0;
        for (int t = 1; t < 400; t++) { term *= (x * x / 4.0) / ((double)t * t); sum += term; if (term < sum * 1e-17) break; }
        lnI0[j] = (float)std::log(sum);
    }
    for (int i = 0; i < kPhiN; i++) {
        const int oct = i / kPhiSteps, st = i % kPhiSteps;
        const double x0 = std::ldexp(1.0 + (double)st / kPhiSteps, kPhiLoExp + oct);    // bin start
        const double xc = std::ldexp(1.0 + (st + 0.5) / kPhiSteps, kPhiLoExp + oct);    // bin centre
        phi[i] = (float)(-std::log(std::tanh(xc / 2.0)));
        if (x0 >= (double)kPhiXHi) phi[i] = 0.0f;                                     // phi0(x > 10) = 0
        if (x0 <= (double)kPhiXLo) phi[i] = 10.0f;                                    // phi0(x < 9.08e-5) = 10 (the clamp's own bin and everything below it)
    }
    bool ok = up(&h->d_row_ptr, rp.data(), rp.size() * 2) && up(&h->d_col_idx, ci.data(), ci.size() * 2) &&
              up(&h->d_col_ptr, cp.data(), cp.size() * 2) && up(&h->d_col_edge, ce.data(), ce.size() * 2) &&
              up(&h->d_lnI0, lnI0.data(), lnI0.size() * 4) && up(&h->d_phi, phi.data(), phi.size() * 4);
    ok = ok && hipMalloc((void **)&h->d_llr_hist, sizeof(uint16_t) * (size_t)nstreams * 2 * c.bits_per_frame()) == hipSuccess;
    ok = ok && hipMalloc((void **)&h->d_fsm, sizeof(FsmState) * (size_t)nstreams) == hipSuccess;
    h->layout = make_decoder_layout(c);
    if (h->layout.ok)
        ok = ok && up(&h->d_rcol, h->layout.rcol.data(), h->layout.rcol.size() * 2) && up(&h->d_vedge, h->layout.vedge.data(), h->layout.vedge.size() * 2) &&
             up(&h->d_vsrc, h->layout.vsrc.data(), h->layout.vsrc.size() * 2);
    if (h->layout.ok && ok) {
        // CRC-16/CCITT-FALSE (the serial loop of decode_fast_kernel; fsk_ldpc.cpp: crc16_ccitt) over the k/8 payload bytes as an affine map of the bits
        const int nbytes = c.k / 8;
        auto crc_of = [&](const std::vector<uint8_t> &bytes, uint16_t init) {
            uint16_t crc = init;
            for (int i = 0; i < nbytes; i++) {
                uint8_t x = (uint8_t)(crc >> 8) ^ bytes[(size_t)i];
                x ^= x >> 4;
                crc = (uint16_t)((crc << 8) ^ ((uint16_t)x << 12) ^ ((uint16_t)x << 5) ^ (uint16_t)x);
            }
            return crc;
        };
        std::vector<uint8_t> msg((size_t)nbytes, 0);
        h->crc0 = crc_of(msg, 0xFFFF);
        std::vector<uint16_t> vcrc((size_t)kFastVars, 0);
        for (int q = 0; q < kFastVars; q++) {
            const int v = h->layout.vsrc[(size_t)q];
            if (v == 0xFFFF || v >= 8 * nbytes) continue;
            msg[(size_t)(v / 8)] = (uint8_t)(0x80u >> (v % 8));
            vcrc[(size_t)q] = crc_of(msg, 0);
            msg[(size_t)(v / 8)] = 0;
        }
        ok = up(&h->d_vcrc, vcrc.data(), vcrc.size() * 2);
    }
    if (!ok) { pirip_hip_ldpc_destroy(h); return PIRIP_ERR_NOMEM; }
    if (h->layout.ok) {
        // decode_fast_kernel folds its phi table's LDS address into a literal, which is right while its dynamic LDS starts at 0, i.e. while the
        // kernel has no static LDS: checked here once per handle instead of trusted (a toolchain change or a __shared__ added to the file
        // would otherwise abort the GPU context at the first decode)
        hipFuncAttributes fa{};
        const void *fn = h->fast_deg() == 6 ? (const void *)decode_fast_kernel<4, 6, kFastColDeg> : (const void *)decode_fast_kernel<4, kFastRowDeg, kFastColDeg>;
        if (hipFuncGetAttributes(&fa, fn) == hipSuccess) h->fast_static_lds = (int)fa.sharedSizeBytes;
        (void)hipGetLastError();
    }
    uint32_t uw = 0;
    for (int i = 0; i < kUwBits; i++) uw |= (uint32_t)(c.uw[i] & 1) << (31 - i);
    int max_row_deg = 0;
    for (int i = 0; i < c.m; i++) max_row_deg = std::max(max_row_deg, (int)(c.row_ptr[i + 1] - c.row_ptr[i]));
    h->dev = LdpcDev{c.n, c.k, c.m, (int)c.col_idx.size(), c.max_iter, c.uw_thresh1, c.uw_thresh2, c.bad_uw_thresh, M, Nsym, Nbits,
                     c.bits_per_frame(), max_row_deg, uw, h->d_row_ptr, h->d_col_idx, h->d_col_ptr, h->d_col_edge, h->d_lnI0, h->d_phi, c.llr_map};
    const int rc = pirip_hip_ldpc_reset(h, nullptr);
    if (rc != PIRIP_OK) { pirip_hip_ldpc_destroy(h); return rc; }
    *out = h;
    return PIRIP_OK;
}

int pirip_hip_ldpc_destroy(pirip_hip_ldpc *h)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    (void)bind_dev(h);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {h->d_rcol, h->d_vedge, h->d_vsrc, h->d_vcrc, h->d_row_ptr, h->d_col_idx, h->d_col_ptr, h->d_col_edge, h->d_lnI0, h->d_phi, h->d_llr_hist, h->d_fsm, h->d_llr_all,
                    h->d_words, h->d_best, h->d_jobs, h->d_njobs, h->d_filt_work, h->d_h_filt, h->d_h_status, h->d_h_payload, h->d_h_info,
                    h->d_dd_llr, h->d_dd_bits, h->d_dd_ip};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (hipStream_t st : h->side) if (st) (void)hipStreamDestroy(st);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_gfork) (void)hipEventDestroy(h->ev_gfork);
    for (hipEvent_t e : h->ev_join) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_mid) if (e) (void)hipEventDestroy(e);
    delete h;
    return PIRIP_OK;
}

int pirip_hip_ldpc_get_info(const pirip_hip_ldpc *h, pirip_ldpc_info *info)
{
    if (!h || !info) return PIRIP_ERR_BAD_ARG;
    std::memset(info, 0, sizeof(*info));
    info->n = h->code.n; info->k = h->code.k; info->bits_per_frame = h->code.bits_per_frame(); info->data_bytes = h->code.data_bytes();
    info->nbits_per_call = h->dev.Nbits; info->max_iter = h->code.max_iter; info->nstreams = h->nstreams;
    std::strncpy(info->name, h->code.name.c_str(), sizeof(info->name) - 1);
    return PIRIP_OK;
}

int pirip_hip_ldpc_reset(pirip_hip_ldpc *h, void *hip_stream)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    LCHK(hipMemsetAsync(h->d_llr_hist, 0, sizeof(uint16_t) * (size_t)h->nstreams * 2 * h->dev.bpf, st));
    LCHK(hipMemsetAsync(h->d_fsm, 0, sizeof(FsmState) * (size_t)h->nstreams, st));
    return PIRIP_OK;
}

namespace {
struct BatchDims { int nbits_total, nwords, max_jobs; size_t llr_stride; };
BatchDims batch_dims(const LdpcDev &c, int ncalls)
{
    BatchDims b;
    b.nbits_total = 2 * c.bpf + ncalls * c.Nbits;
    b.nwords = (b.nbits_total + 31) / 32 + 1;
    b.max_jobs = (ncalls * c.Nbits) / c.bpf + 2;
    b.llr_stride = (size_t)b.nbits_total;
    return b;
}
int ensure_work(pirip_hip_ldpc *h, int ncalls, hipStream_t st);
int stages_after_llr(pirip_hip_ldpc *h, const int32_t *d_ncalls, int ncalls, uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, hipStream_t st,
                     int s0 = 0, int n = -1, bool beside_demod = false, hipStream_t sdec = nullptr, hipEvent_t ev = nullptr);
hipStream_t side_stream(pirip_hip_ldpc *h, int slot);
bool side_events(pirip_hip_ldpc *h, int n);
}  // namespace

int pirip_hip_ldpc_rx_batch(pirip_hip_ldpc *h, const float *d_rx_filt, size_t filt_stride, const int32_t *d_ncalls, int ncalls,
                            uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, void *hip_stream)
{
    if (!h || !d_rx_filt || !d_status || !d_payload || !d_info || ncalls < 0) return PIRIP_ERR_BAD_ARG;
    if (ncalls == 0) return PIRIP_OK;
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const LdpcDev &c = h->dev;
    const BatchDims bd = batch_dims(c, ncalls);
    int rc = ensure_work(h, ncalls, st);
    if (rc != PIRIP_OK) return rc;
    const bool fused_words = (2 * c.bpf) % 32 == 0;        // every LLR tile then covers whole hard-decision words
    LCHK(launch_llr<h16>(c, dim3((ncalls + kLlrTile - 1) / kLlrTile, h->nstreams), st, d_rx_filt, filt_stride, d_ncalls, ncalls, h->d_llr_all, bd.llr_stride,
                         h->d_llr_hist, fused_words ? h->d_words : (uint32_t *)nullptr, bd.nwords));
    if (!fused_words)
        hipLaunchKernelGGL(hard_kernel, dim3((bd.nwords + 255) / 256, h->nstreams), dim3(256), 0, st, h->d_llr_all, bd.llr_stride, bd.nbits_total, h->d_words, bd.nwords);
    return stages_after_llr(h, d_ncalls, ncalls, d_status, d_payload, d_info, st);
}

// The whole FSK_LDPC receive chain of one batch (include/pirip_hip.h section E): IQ -> status / payload / info records. Where the
// demodulator's instance can (demod_wave_soft_capable) the bit LLRs and their hard-decision words are written by the demodulator
// itself -- no soft magnitudes in HBM, no LLR kernel; otherwise magnitudes go through a work buffer and pirip_hip_ldpc_rx_batch.
int pirip_hip_fsk_ldpc_rx_batch(pirip_hip_demod *dem, pirip_hip_ldpc *h, const void *d_in, size_t in_stride_bytes, int64_t nsamp,
                                uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, float *d_stats, size_t stats_stride,
                                int32_t *d_nframes, int64_t *d_consumed, int64_t max_frames, void *hip_stream)
{
    if (!dem || !h || !d_in || !d_status || !d_payload || !d_info || !d_nframes || nsamp < 0 || max_frames <= 0 || max_frames > (1 << 24)) return PIRIP_ERR_BAD_ARG;
    int M = 0, Nsym = 0, ns = 0, dev = 0;
    if (demod_handle_shape(dem, &M, &Nsym, &ns, &dev) != PIRIP_OK) return PIRIP_ERR_BAD_ARG;
    const LdpcDev &c = h->dev;
    if (M != c.M || Nsym != c.Nsym || ns != h->nstreams || dev != h->device) return PIRIP_ERR_BAD_ARG;   // the two handles describe the same streams
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const int ncalls = (int)max_frames;
    const BatchDims bd = batch_dims(c, ncalls);
    int rc = ensure_work(h, ncalls, st);
    if (rc != PIRIP_OK) return rc;
    if ((2 * c.bpf) % 32 == 0 && demod_soft_capable(dem, nsamp)) {
        const SoftOut so{h->d_llr_all, bd.llr_stride, h->d_words, (size_t)bd.nwords, h->d_lnI0, 2 * c.bpf, c.llr_map};
        // the chain of receivers [s0, s0 + n) on stream sg
        // (sdec / ev: the decode on another stream, ordered behind the range's earlier stages by the event)
        auto run_range = [&](int s0, int n, hipStream_t sg, bool beside = false, hipStream_t sdec = nullptr, hipEvent_t ev = nullptr) -> int {
            if (n <= 0) return PIRIP_OK;
            LCHK(hipMemsetAsync(h->d_words + (size_t)s0 * bd.nwords, 0, sizeof(uint32_t) * (size_t)n * bd.nwords, sg));
            hipLaunchKernelGGL(hist_prepare_kernel, dim3((2 * c.bpf + 255) / 256, n), dim3(256), 0, sg, c.bpf, h->d_llr_hist + (size_t)s0 * (size_t)(2 * c.bpf),
                               h->d_llr_all + (size_t)s0 * bd.llr_stride, bd.llr_stride, h->d_words + (size_t)s0 * bd.nwords, bd.nwords);
            LCHK(hipGetLastError());
            const int r = demod_batch_soft(dem, d_in, in_stride_bytes, nsamp, so, d_stats, stats_stride, d_nframes, d_consumed, max_frames, sg, s0, n);
            if (r != PIRIP_OK) return r;
            return stages_after_llr(h, d_nframes, ncalls, d_status, d_payload, d_info, sg, s0, n, beside, sdec, ev);
        };
        h->last_path_fused = 1;
        // Many streams: two ranges (5/8 and 3/8 of them) on two internal HIP streams, forked from and joined back into the caller's. The
        // FSK_LDPC stages are bound by the LDS pipe and the demodulator by VALU issue: the first range's decode (high priority) runs beside
        // the second range's demodulator instead of after the whole batch's (config 4: 26.9 -> 25.2 ms at 3.5 dB). Same kernels on the same
        // per-stream data: the records do not depend on the split. PIRIP_CHAIN_SPLIT_MIN=<streams> (read when the handle is created)
        // moves the threshold (0: never split).
        // (up to four ranges: the last one on slot 0 at low priority, the ones before it on slots 1, 10, 11 at high priority)
        constexpr int kRangeSlot[4] = {1, 10, 11, 0};
        int nr = h->n_bounds + 1;
        if (!(h->split_min > 0 && h->nstreams >= h->split_min && h->nstreams >= 2 && side_events(h, pirip_hip_ldpc::kSideSlots))) nr = 1;
        // Which decoder for the first range, and where the ranges end. The persistent decoder takes whole CUs, so beside the second range's
        // demodulator it only fills that kernel's tail. decode_fast_kernel's workgroups (40 KB, 128 VGPR) fit beside two demodulator workgroups
        // and, at the first range's high priority, their LDS-bound waves issue between the VALU-bound ones: real overlap (config 4: 24.5 ->
        // 23.7 ms at 3.5 dB) -- PROVIDED no demodulator workgroup of the second range is still waiting for a place when that decode
        // starts: a high-priority decode would take the places first and the second range's demodulator would finish after it (measured:
        // 28 ms). With halves A = B, a chip that holds Cs streams of the demodulator at a time and first-range workgroups placed first, A's
        // last round starts at (k - 1) rounds, k = ceil(A / Cs), and when it ends k Cs - A + Cs places have gone to B: overlap is chosen when
        // that covers B, i.e. nstreams <= (k + 1) Cs (8192 streams on 256 CUs x 12: yes; 12288: no -> persistent decoder, 5/8 + 3/8 as before).
        int overlap = h->overlap_decoder_fast;
        int first_end = -1;
        if (overlap < 0) {
            overlap = 0;
            const int64_t cs = (int64_t)demod_streams_per_cu(dem) * h->num_cu;
            const int64_t half = ((int64_t)h->nstreams / 2 + 3) & ~(int64_t)3;
            if (nr == 2 && cs > 0 && !h->in_group && h->layout.ok && h->decoder_pref == kDecAuto && h->fast_static_lds == 0) {
                const int64_t k = (half + cs - 1) / cs;
                if ((int64_t)h->nstreams <= (k + 1) * cs) { overlap = 1; first_end = (int)half; }
            }
        }
        // "fast-low": the earlier ranges' decode (the small decoder) goes to slot 0, the lowest priority, and the last range to the middle one:
        // a decoder workgroup then takes a place on a CU only when no demodulator workgroup is waiting for it -- it fills the last
        // demodulator round's gaps (two demodulator workgroups + decoder waves on a CU) and never delays the demodulator itself
        const bool dec_low = overlap == 2 && nr > 1;
        hipStream_t sr[4] = {nullptr, nullptr, nullptr, nullptr}, sdec = nullptr;
        int slot[4] = {0, 0, 0, 0}, end[4] = {0, 0, 0, 0};
        for (int i = 0; i < nr && nr > 1; i++) {
            slot[i] = i == nr - 1 ? (dec_low ? pirip_hip_ldpc::kMidSlot : kRangeSlot[3]) : kRangeSlot[i];
            sr[i] = side_stream(h, slot[i]);
            if (!sr[i]) nr = 1;
        }
        if (dec_low && nr > 1 && !(sdec = side_stream(h, 0))) nr = 1;
        if (nr == 1) return run_range(0, h->nstreams, st);
        for (int i = 0; i < nr; i++) {
            end[i] = i == nr - 1 ? h->nstreams : (int)(((int64_t)h->nstreams * h->split_bounds[i] / 8 + 3) & ~3);
            if (end[i] > h->nstreams) end[i] = h->nstreams;
        }
        if (nr == 2 && first_end > 0) end[0] = first_end;
        if (nr == 2 && end[0] >= h->nstreams) end[0] = h->nstreams / 2;
        // fork: nothing has been launched on the side streams if one of these fails
        LCHK(hipEventRecord(h->ev_fork, st));
        for (int i = 0; i < nr; i++) LCHK(hipStreamWaitEvent(sr[i], h->ev_fork, 0));
        if (sdec) LCHK(hipStreamWaitEvent(sdec, h->ev_fork, 0));
        rc = PIRIP_OK;
        for (int i = 0; i < nr && rc == PIRIP_OK; i++) {
            const int s0 = i ? end[i - 1] : 0;
            const bool beside = overlap != 0 && i < nr - 1;              // (every range but the last decodes beside a demodulator: the small decoder)
            rc = h->test_fail_range == i ? PIRIP_ERR_HIP : run_range(s0, end[i] - s0, sr[i], beside, beside ? sdec : nullptr, beside && sdec ? h->ev_mid[i] : nullptr);
        }
        // join on EVERY path: whatever the ranges did launch is ordered before the caller's next work on its stream
        hipError_t jerr = hipSuccess;
        for (int i = 0; i < nr + (sdec ? 1 : 0); i++) {
            const int sl = i < nr ? slot[i] : 0;
            hipError_t e = hipEventRecord(h->ev_join[sl], i < nr ? sr[i] : sdec);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, h->ev_join[sl], 0);
            if (e != hipSuccess) jerr = e;
        }
        if (rc != PIRIP_OK) return rc;
        LCHK(jerr);
        return PIRIP_OK;
    }
    // no fused instance for this shape (general kernel, fsk_demod -p 24, a code whose window is not a whole number of words)
    h->last_path_fused = 0;
    const size_t per = (size_t)c.M * c.Nsym;
    if ((size_t)ncalls > h->filt_cap) {
        LCHK(hipStreamSynchronize(st));
        if (h->d_filt_work) (void)hipFree(h->d_filt_work);
        h->d_filt_work = nullptr; h->filt_cap = 0;
        LCHK(hipMalloc((void **)&h->d_filt_work, sizeof(float) * (size_t)h->nstreams * ncalls * per));
        h->filt_cap = (size_t)ncalls;
    }
    rc = pirip_hip_demod_batch(dem, d_in, in_stride_bytes, nsamp, nullptr, 0, h->d_filt_work, (size_t)ncalls * per, d_stats, stats_stride, d_nframes, d_consumed,
                               max_frames, hip_stream);
    if (rc != PIRIP_OK) return rc;
    return pirip_hip_ldpc_rx_batch(h, h->d_filt_work, (size_t)ncalls * per, d_nframes, ncalls, d_status, d_payload, d_info, hip_stream);
}

int pirip_hip_fsk_ldpc_last_path(const pirip_hip_ldpc *h) { return h ? h->last_path_fused : PIRIP_ERR_BAD_ARG; }

// several groups of streams, each on its own prioritised HIP stream (header): a fork / join around pirip_hip_fsk_ldpc_rx_batch
int pirip_hip_fsk_ldpc_rx_batch_groups(const pirip_chain_group *groups, int ngroups, size_t in_stride_bytes, int64_t nsamp, size_t stats_stride,
                                       int64_t max_frames, void *hip_stream)
{
    constexpr int kMaxGroups = 8;
    if (!groups || ngroups < 1 || ngroups > kMaxGroups) return PIRIP_ERR_BAD_ARG;
    for (int g = 0; g < ngroups; g++) {
        if (!groups[g].dem || !groups[g].ldpc) return PIRIP_ERR_BAD_ARG;
        if (groups[g].ldpc->device != groups[0].ldpc->device) return PIRIP_ERR_BAD_ARG;
    }
    if (ngroups == 1)
        return pirip_hip_fsk_ldpc_rx_batch(groups[0].dem, groups[0].ldpc, groups[0].d_in, in_stride_bytes, nsamp, groups[0].d_status, groups[0].d_payload,
                                           groups[0].d_info, groups[0].d_stats, stats_stride, groups[0].d_nframes, groups[0].d_consumed, max_frames, hip_stream);
    pirip_hip_ldpc *h = groups[0].ldpc;                            // (whose side streams / events carry the groups, and where LCHK records a HIP error)
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    if (!side_events(h, pirip_hip_ldpc::kGroupSlot0 + ngroups)) return PIRIP_ERR_HIP;
    for (int g = 0; g < ngroups; g++) if (!side_stream(h, pirip_hip_ldpc::kGroupSlot0 + g)) return PIRIP_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;
    LCHK(hipEventRecord(h->ev_gfork, st));
    int rc = PIRIP_OK;
    hipError_t jerr = hipSuccess;
    for (int g = 0; g < ngroups && rc == PIRIP_OK; g++) {
        const int slot = pirip_hip_ldpc::kGroupSlot0 + (g == ngroups - 1 ? 0 : 1 + g);      // the last group at low priority, the others high
        hipStream_t sg = side_stream(h, slot);
        if (hipStreamWaitEvent(sg, h->ev_gfork, 0) != hipSuccess) { rc = PIRIP_ERR_HIP; break; }
        // (a group that splits again inside does so on its own handle's slots 0 / 1: no two pieces of work share a side stream)
        groups[g].ldpc->in_group = 1;
        rc = pirip_hip_fsk_ldpc_rx_batch(groups[g].dem, groups[g].ldpc, groups[g].d_in, in_stride_bytes, nsamp, groups[g].d_status, groups[g].d_payload,
                                         groups[g].d_info, groups[g].d_stats, stats_stride, groups[g].d_nframes, groups[g].d_consumed, max_frames, (void *)sg);
        groups[g].ldpc->in_group = 0;
        // join this group whether or not it succeeded
        hipError_t e = hipEventRecord(h->ev_join[slot], sg);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, h->ev_join[slot], 0);
        if (e != hipSuccess) jerr = e;
    }
    if (rc != PIRIP_OK) return rc;
    LCHK(jerr);
    return PIRIP_OK;
}

}  // extern "C"

namespace {
int ensure_work(pirip_hip_ldpc *h, int ncalls, hipStream_t st)
{
    const LdpcDev &c = h->dev;
    const size_t ns = (size_t)h->nstreams;
    const BatchDims bd = batch_dims(c, ncalls);
    const int nbits_total = bd.nbits_total, nwords = bd.nwords, max_jobs = bd.max_jobs;
    if ((size_t)ncalls > h->cap_calls) {
        LCHK(hipStreamSynchronize(st));
        void *olds[] = {h->d_llr_all, h->d_words, h->d_best, h->d_jobs, h->d_njobs};
        for (void *p : olds) if (p) (void)hipFree(p);
        h->d_llr_all = nullptr; h->d_words = nullptr; h->d_best = nullptr; h->d_jobs = nullptr; h->d_njobs = nullptr; h->cap_calls = 0;
        LCHK(hipMalloc((void **)&h->d_llr_all, sizeof(uint16_t) * ns * nbits_total + 16));
        LCHK(hipMalloc((void **)&h->d_words, sizeof(uint32_t) * ns * nwords));
        LCHK(hipMalloc((void **)&h->d_best, sizeof(uint32_t) * ns * ncalls));
        LCHK(hipMalloc((void **)&h->d_jobs, sizeof(int32_t) * ns * max_jobs * 2));
        LCHK(hipMalloc((void **)&h->d_njobs, sizeof(int32_t) * ns));
        h->cap_calls = (size_t)ncalls;
    }
    return PIRIP_OK;
}

// s0 / n (n < 0: all): receivers [s0, s0 + n) only; d_ncalls / d_status / d_payload / d_info are the caller's arrays of receiver 0
// sdec / ev: the decode (and what follows it) on stream sdec, ordered behind the unique-word search and the sync logic by the event
int stages_after_llr(pirip_hip_ldpc *h, const int32_t *d_ncalls, int ncalls, uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, hipStream_t st, int s0, int n, bool beside_demod,
                     hipStream_t sdec, hipEvent_t ev)
{
    const LdpcDev &c = h->dev;
    if (n < 0) { s0 = 0; n = h->nstreams; }
    if (n == 0) return PIRIP_OK;
    const size_t ns = (size_t)n, z = (size_t)s0;
    const BatchDims bd = batch_dims(c, ncalls);
    const int nbits_total = bd.nbits_total, nwords = bd.nwords, max_jobs = bd.max_jobs;
    const size_t llr_stride = bd.llr_stride;
    // this range's slices of the per-receiver arrays
    uint32_t *words = h->d_words + z * nwords, *best = h->d_best + z * ncalls;
    int32_t *jobs = h->d_jobs + z * max_jobs * 2, *njobs = h->d_njobs + z;
    uint16_t *llr_all = h->d_llr_all + z * llr_stride, *llr_hist = h->d_llr_hist + z * (size_t)(2 * c.bpf);
    FsmState *fsm = h->d_fsm + z;
    if (d_ncalls) d_ncalls += z;
    d_status += z * ncalls; d_payload += z * ncalls * (size_t)(c.k / 8); d_info += z * ncalls * kInfoPerCall;
    LCHK(hipMemsetAsync(d_payload, 0, ns * ncalls * (size_t)(c.k / 8), st));
    {
        const int K = c.bpf / c.Nbits;
        const size_t lds = sizeof(uint32_t) * ((size_t)(((kUwCalls + K + 1) * c.Nbits + 31) / 32 + 4) + 2 * (size_t)(kUwCalls + K + 1));
        if (lds > 64 * 1024) return PIRIP_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(uwbest_kernel, dim3((ncalls + kUwCalls - 1) / kUwCalls, n), dim3(256), lds, st, c, ncalls, words, nwords, nbits_total, best);
    }
    hipLaunchKernelGGL(fsm_kernel, dim3((n + 63) / 64), dim3(64), 0, st, c, n, ncalls, d_ncalls, words, nwords, best, nbits_total, fsm,
                       d_status, d_info, jobs, njobs, max_jobs);
    LCHK(hipGetLastError());
    if (sdec && ev) { LCHK(hipEventRecord(ev, st)); LCHK(hipStreamWaitEvent(sdec, ev, 0)); st = sdec; }
    const int rc = launch_decode(h, max_jobs, n, jobs, njobs, llr_all, llr_stride, 0, d_status, ncalls, d_payload,
                                 d_info, nullptr, nullptr, st, beside_demod);
    if (rc != PIRIP_OK) return rc;
    hipLaunchKernelGGL(save_hist_kernel, dim3((2 * c.bpf + 255) / 256, n), dim3(256), 0, st, llr_all, llr_stride, ncalls, d_ncalls, c.Nbits, c.bpf, llr_hist);
    LCHK(hipGetLastError());
    return PIRIP_OK;
}

// this handle's internal HIP streams for work that runs beside the caller's stream (made on first use, kept until destroy): slots 0 and
// kGroupSlot0 at the lowest priority, the others at the highest; nullptr if they cannot be made
hipStream_t side_stream(pirip_hip_ldpc *h, int slot)
{
    if (slot < 0 || slot >= pirip_hip_ldpc::kSideSlots) return nullptr;
    if (!h->side[slot]) {
        int lo = 0, hi = 0;                                         // (numerically: greatest priority = the smaller number)
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return nullptr;
        const bool low = slot == 0 || slot == pirip_hip_ldpc::kGroupSlot0;
        const int prio = low ? lo : slot == pirip_hip_ldpc::kMidSlot ? (lo + hi) / 2 : hi;
        if (hipStreamCreateWithPriority(&h->side[slot], hipStreamNonBlocking, prio) != hipSuccess) { h->side[slot] = nullptr; return nullptr; }
    }
    return h->side[slot];
}
// the fork events and the join events of slots [0, n): made once per handle
bool side_events(pirip_hip_ldpc *h, int n)
{
    if (n > pirip_hip_ldpc::kSideSlots) return false;
    if (!h->ev_fork && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) { h->ev_fork = nullptr; return false; }
    if (!h->ev_gfork && hipEventCreateWithFlags(&h->ev_gfork, hipEventDisableTiming) != hipSuccess) { h->ev_gfork = nullptr; return false; }
    for (int i = 0; i < n; i++)
        if (!h->ev_join[i] && hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess) { h->ev_join[i] = nullptr; return false; }
    for (hipEvent_t &e : h->ev_mid)
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return false; }
    return true;
}
}  // namespace

extern "C" {

int pirip_hip_ldpc_rx_host(pirip_hip_ldpc *h, const float *rx_filt, int ncalls, uint8_t *status, uint8_t *payload, int32_t *info)
{
    if (!h || (!rx_filt && ncalls > 0) || ncalls < 0 || !status || !payload || !info) return PIRIP_ERR_BAD_ARG;
    if (h->nstreams != 1) return PIRIP_ERR_BAD_ARG;
    if (ncalls == 0) return PIRIP_OK;
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    const LdpcDev &c = h->dev;
    const size_t per = (size_t)c.M * c.Nsym, nb = (size_t)(c.k / 8);
    if ((size_t)ncalls > h->h_cap) {
        void *olds[] = {h->d_h_filt, h->d_h_status, h->d_h_payload, h->d_h_info};
        for (void *p : olds) if (p) (void)hipFree(p);
        h->d_h_filt = nullptr; h->d_h_status = nullptr; h->d_h_payload = nullptr; h->d_h_info = nullptr; h->h_cap = 0;
        LCHK(hipMalloc((void **)&h->d_h_filt, sizeof(float) * per * ncalls));
        LCHK(hipMalloc((void **)&h->d_h_status, (size_t)ncalls));
        LCHK(hipMalloc((void **)&h->d_h_payload, nb * ncalls));
        LCHK(hipMalloc((void **)&h->d_h_info, sizeof(int32_t) * kInfoPerCall * ncalls));
        h->h_cap = (size_t)ncalls;
    }
    LCHK(hipMemcpy(h->d_h_filt, rx_filt, sizeof(float) * per * ncalls, hipMemcpyHostToDevice));
    const int rc = pirip_hip_ldpc_rx_batch(h, h->d_h_filt, 0, nullptr, ncalls, h->d_h_status, h->d_h_payload, h->d_h_info, nullptr);
    if (rc != PIRIP_OK) return rc;
    LCHK(hipDeviceSynchronize());
    LCHK(hipMemcpy(status, h->d_h_status, (size_t)ncalls, hipMemcpyDeviceToHost));
    LCHK(hipMemcpy(payload, h->d_h_payload, nb * ncalls, hipMemcpyDeviceToHost));
    LCHK(hipMemcpy(info, h->d_h_info, sizeof(int32_t) * kInfoPerCall * ncalls, hipMemcpyDeviceToHost));
    return PIRIP_OK;
}

int pirip_hip_ldpc_decode_llr(pirip_hip_ldpc *h, const float *d_llr, int ncw, uint8_t *d_bits, int32_t *d_iter_pcc, void *hip_stream)
{
    if (!h || !d_llr || !d_bits || !d_iter_pcc || ncw < 0) return PIRIP_ERR_BAD_ARG;
    if (ncw == 0) return PIRIP_OK;
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t nll = (size_t)ncw * h->dev.n;
    if (nll > h->dd_cap) {
        LCHK(hipStreamSynchronize(st));
        if (h->d_dd_llr) (void)hipFree(h->d_dd_llr);
        h->d_dd_llr = nullptr; h->dd_cap = 0;
        LCHK(hipMalloc((void **)&h->d_dd_llr, sizeof(uint16_t) * nll));
        h->dd_cap = nll;
    }
    hipLaunchKernelGGL(f32_to_h16_kernel, dim3((unsigned)((nll + 255) / 256)), dim3(256), 0, st, d_llr, h->d_dd_llr, nll);   // the decoder's input format
    LCHK(hipGetLastError());
    return launch_decode(h, ncw, 1, nullptr, nullptr, h->d_dd_llr, 0, 1, nullptr, 0, nullptr, nullptr, d_bits, d_iter_pcc, st);
}

int pirip_hip_ldpc_llr(pirip_hip_ldpc *h, const float *d_rx_filt, int ncalls, float *d_llr, void *hip_stream)
{
    if (!h || !d_rx_filt || !d_llr || ncalls < 0) return PIRIP_ERR_BAD_ARG;
    if (ncalls == 0) return PIRIP_OK;
    if (!bind_dev(h)) return PIRIP_ERR_NO_DEVICE;
    // one pseudo-stream whose history slot is skipped: write straight to d_llr (offset so that "2*bpf + call*Nbits" lands at call*Nbits)
    const LdpcDev &c = h->dev;
    LCHK(launch_llr<float>(c, dim3((ncalls + kLlrTile - 1) / kLlrTile, 1), (hipStream_t)hip_stream, d_rx_filt, (size_t)0, (const int32_t *)nullptr, ncalls,
                           d_llr - 2 * c.bpf, (size_t)0, (const h16 *)nullptr, (uint32_t *)nullptr, 0));
    LCHK(hipGetLastError());
    return PIRIP_OK;
}

}  // extern "C"
