// pirip_amd/csrc/decim_kernels.hip -- csdr front end on the GPU (include/pirip_hip.h section B, D).
//
//   csdr convert_u8_f | csdr fir_decimate_cc D [tbw] | csdr convert_f_s16
//   (/root/reference/README.md:109,162; arithmetic UPSTREAM-RECALLED from ha7ilm/csdr libcsdr.c:
//    convert_u8_f, fir_decimate_cc, convert_f_s16)
//
// What was measured on the way (64 streams x 45e6 samples, clocks warm, fraction of the 8 TB/s HBM spec;
// the read-only ceiling measured with tools/hbm_read_ceiling.hip is 6.3 TB/s = 79 %):
//   window loads issued one chunk at a time behind per-chunk bounds branches ........ 31 %  (round-1 first cut)
//   all 16-byte loads of a tile in flight together (buffer descriptor bounds check) .. 42 %
//   sample reads kept as aligned 16-bit LDS reads (the compiler had merged them into
//   unaligned 8/16-byte reads: SQ_LDS_UNALIGNED_STALL = 3/4 of the LDS-active cycles),
//   two-fma exact u8->float (6 instead of 7 VALU instructions per tap) .............. 54-56 %  <- this kernel
// and did NOT help (all removed again): converting each sample once with float staging in LDS (15-20 %:
// occupancy), reading the next tile into registers while filtering (+-0), two adjacent outputs per lane sharing
// the conversion (14 % fewer VALU instructions, half the LDS traffic, conflict-free, but half the waves: +-0),
// 79 taps fully unrolled with the 40 distinct values in SGPRs (spills to v_readlane) or VGPRs (122 VGPRs: -2 %),
// and (round 2) a neighbour-sharing variant for D = 45 / L = 79: each lane converts only its own 45 samples, keeps the first 34
// in registers and takes taps 45..78 from the next lane through DPP (wave_shl:1; 63 outputs per wave) -- 15 % fewer VALU
// instructions and 43 % fewer LDS reads, bit-exact, but 95-176 VGPRs instead of 32: 3.55-4.36 ms against this kernel's 3.47.
// (On the way: two __builtin_amdgcn_update_dpp calls on the halves of one register pair compiled, with hipcc 7.2, into ONE
// DPP move feeding both halves -- found by the bit-exact s16 test; inline-asm v_mov_b32_dpp is the workaround.)
// Under sustained load this stage runs the board at its 1400 W power cap (rocm-smi: sclk ~1.79 GHz instead of
// 2.4 GHz; the demodulator draws ~290 W at 2.4 GHz), which is where the 95 %-VALU-busy kernel lands at 1.36 ms
// instead of the ~1.0 ms its instruction count would need at full clock. The two-outputs-per-lane variant was
// re-measured in that regime (6 s runs): 51 % against 54.5 %.
// fused into one kernel: u8 IQ in (2 B per sample from HBM), decimated complex out (s16 or
// f32, 4-8 B per D input samples). Direct form, real taps, no zero pre-history:
//     y[k] = sum_{t=0}^{L-1} h[t] * x[k*D + t]       (I and Q separately)
// Each output is accumulated in ascending tap order with separate multiply and add
// (-ffp-contract=off), which is the scalar csdr loop's float32 result bit for bit; the
// parallelism is across outputs. A workgroup stages the u8 span of its output tile in LDS
// with 16-byte coalesced loads (each input byte is read from HBM once per tile; tiles overlap
// by L-D samples), the taps and the 256-entry u8->float table sit in LDS too.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pirip_hip.h"
#include "fsk_plan.hpp"

using namespace pirip;

namespace {

constexpr int kThreads = 256;
constexpr int kGroup = 6;        // 6 x 256 x 16 B = 24 KiB of window per load group (one group at D = 45)

struct DecimArgs {
    const uint8_t *in; size_t in_stride; int64_t n_in;
    void *out; size_t out_stride; int64_t n_out;
    const float *taps; const float *lut;
    int D, L, tile, out_s16;
    int arith;                 // 1: convert with the exact two-fma formula instead of the LDS look-up table
    int tpw, mode;             // tiles per workgroup; arithmetic of the tap loop (kDecimExact / kDecimFma / kDecimFmaRaw)
    float c_hi, c_lo;
    float tap_sum;             // kDecimFmaRaw: sum of the taps (float of the double sum)
};
// Tap-loop arithmetic. kDecimExact (default): one multiply and one add per tap and component in ascending tap order -- the scalar csdr
// loop's float32 result bit for bit. The other two are OPT-IN measurements (PIRIP_DECIM_FMA=1 / 2): upstream csdr is
// built -O3 -ffast-math [UPSTREAM-RECALLED], so the shipped binary's summation order is its vectoriser's, not the scalar loop's, and
// which float32 result "the reference" produces is a property of that build:
//   kDecimFma     the same conversion, acc = fma(y, h, acc): one v_pk_fma_f32 per tap for the (I, Q) pair, same tap order
//   kDecimFmaRaw  the affine u8 map pulled out of the sum: acc = fma(byte, h, acc), y = acc / 127.5 - sum(h) at the end -- half the
//                 VALU instructions of the exact loop
constexpr int kDecimExact = 0, kDecimFma = 1, kDecimFmaRaw = 2;

// One 16-bit LDS read, kept as such: left to itself the compiler merges the per-tap reads of a lane into
// 8/16-byte ds_reads at 2-byte alignment, and unaligned wide LDS reads stall the LDS pipe (PMC:
// SQ_LDS_UNALIGNED_STALL was 3/4 of the LDS-active cycles and the LDS pipe 95 % busy).
__device__ __forceinline__ uint32_t lds_u16(const uint8_t *p)
{
    return *(const volatile __attribute__((address_space(3))) uint16_t *)p;
}

struct TileGeom {
    int64_t k0, gbase;         // first output of the tile; window start relative to the stream base (bytes)
    const uint8_t *wsrc;       // window start (16-byte aligned address)
    int nouts, head, wlen;     // outputs in the tile; bytes between window start and the first needed byte; window bytes
};

__device__ __forceinline__ TileGeom tile_geom(const DecimArgs &a, const uint8_t *src, int64_t tile)
{
    TileGeom g;
    g.k0 = tile * a.tile;
    const int64_t b0 = 2 * g.k0 * a.D;                    // first byte needed
    g.nouts = (int)((a.n_out - g.k0) < a.tile ? (a.n_out - g.k0) : a.tile);
    const int64_t nbytes = 2 * ((int64_t)(g.nouts - 1) * a.D + a.L);
    g.head = (int)((uintptr_t)(src + b0) & 15);           // 16-byte aligned window covering [b0, b0 + nbytes)
    g.wsrc = src + b0 - g.head;
    g.wlen = (int)((g.head + nbytes + 15) & ~(int64_t)15);
    g.gbase = b0 - g.head;
    return g;
}

// Stage a tile's u8 window in LDS. Bounds-checked 16-byte loads through a buffer descriptor on the window: no
// per-chunk branches, so all loads of a group are in flight together (the tile's HBM latency is paid once, not
// once per chunk). The last chunk of a stream may straddle its end: records are rounded up to 16 bytes, which
// never crosses a page; bytes past the end are never used by a valid output.
__device__ __forceinline__ void stage_window(const TileGeom &g, int64_t total, uint8_t *s_x, int tid)
{
    if (g.gbase >= 0) {
        const int64_t left = ((total - g.gbase) + 15) & ~(int64_t)15;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)g.wsrc, 0, (int)(uint32_t)(left > 0x7ffffff0 ? 0x7ffffff0 : left), 0x00020000);
        for (int o0 = 0; o0 < g.wlen; o0 += kGroup * kThreads * 16) {
            uint4 v[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; i++)
                v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o0 + (i * kThreads + tid) * 16, 0, 0));
#pragma unroll
            for (int i = 0; i < kGroup; i++) {
                const int o = o0 + (i * kThreads + tid) * 16;
                if (o < g.wlen) *(uint4 *)(s_x + o) = v[i];
            }
        }
    } else {
        // first tile of a stream whose base is not 16-byte aligned: the window starts before the stream
        for (int o = tid * 16; o < g.wlen; o += kThreads * 16) {
            const int64_t gg = g.gbase + o;
            uint8_t tmp[16];
            for (int q = 0; q < 16; q++) tmp[q] = (gg + q >= 0 && gg + q < total) ? g.wsrc[o + q] : 0;
            uint4 w;
            memcpy(&w, tmp, 16);
            *(uint4 *)(s_x + o) = w;
        }
    }
}

__device__ __forceinline__ void store_out(const DecimArgs &a, int sid, int64_t k, float acci, float accq)
{
    if (a.out_s16) {
        short2 *o = (short2 *)((char *)a.out + (size_t)sid * a.out_stride) + k;
        *o = make_short2((short)(acci * (float)SHRT_MAX), (short)(accq * (float)SHRT_MAX));
    } else {
        float2 *o = (float2 *)((char *)a.out + (size_t)sid * a.out_stride) + k;
        *o = make_float2(acci, accq);
    }
}

// csdr's u8->float, x/127.5 - 1 evaluated in double and rounded to float, as two fused multiply-adds:
// fma(x, c_lo, fma(x, c_hi, -1)) with c_hi = 1/127.5 rounded to a multiple of 2^-22 (so the inner fma is exact for
// every byte value) and c_lo the float remainder; bit-identical to the double formula for all 256 byte values,
// checked at create time. Then one separate multiply and add per component (the scalar csdr loop's arithmetic).
template <int MODE>
__device__ __forceinline__ void tap_mac(const DecimArgs &a, uint32_t w, float h, float &acci, float &accq)
{
    const float xi = (float)(w & 0xffu), xq = (float)(w >> 8);
    if (MODE == kDecimFmaRaw) { acci = __builtin_fmaf(xi, h, acci); accq = __builtin_fmaf(xq, h, accq); return; }
    const float yi = __builtin_fmaf(xi, a.c_lo, __builtin_fmaf(xi, a.c_hi, -1.0f));
    const float yq = __builtin_fmaf(xq, a.c_lo, __builtin_fmaf(xq, a.c_hi, -1.0f));
    if (MODE == kDecimFma) { acci = __builtin_fmaf(yi, h, acci); accq = __builtin_fmaf(yq, h, accq); return; }
    acci += yi * h;
    accq += yq * h;
}
// the tap loop of one output: one 16-bit LDS read per tap, four taps per 16-byte (broadcast) LDS read
template <int MODE>
__device__ __forceinline__ void tap_loop(const DecimArgs &a, const uint8_t *x, const float *s_taps, float &acci, float &accq)
{
    int t = 0;
#pragma unroll 2
    for (; t + 4 <= a.L; t += 4) {
        const float4 h = *(const float4 *)(s_taps + t);
        const uint32_t w0 = lds_u16(x + 2 * t), w1 = lds_u16(x + 2 * t + 2);
        const uint32_t w2 = lds_u16(x + 2 * t + 4), w3 = lds_u16(x + 2 * t + 6);
        tap_mac<MODE>(a, w0, h.x, acci, accq); tap_mac<MODE>(a, w1, h.y, acci, accq);
        tap_mac<MODE>(a, w2, h.z, acci, accq); tap_mac<MODE>(a, w3, h.w, acci, accq);
    }
    for (; t < a.L; t++) tap_mac<MODE>(a, lds_u16(x + 2 * t), s_taps[t], acci, accq);
    if (MODE == kDecimFmaRaw) {
        const float c = a.c_hi + a.c_lo;                     // 1 / 127.5 rounded to float
        acci = __builtin_fmaf(acci, c, -a.tap_sum); accq = __builtin_fmaf(accq, c, -a.tap_sum);
    }
}

// General kernel: any tap count, any alignment, taps and the u8->float table in LDS.
__global__ __launch_bounds__(kThreads) void decim_kernel(DecimArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_taps = (float *)smem;                       // [L] (L rounded up to 4)
    float *s_lut = s_taps + ((a.L + 3) & ~3);            // [256]
    uint8_t *s_x = (uint8_t *)(s_lut + 256);             // staged u8 IQ, 16-B aligned

    const int tid = threadIdx.x;
    const int sid = blockIdx.y;
    for (int i = tid; i < a.L; i += kThreads) s_taps[i] = a.taps[i];
    for (int i = tid; i < 256; i += kThreads) s_lut[i] = a.lut[i];

    const uint8_t *src = a.in + (size_t)sid * a.in_stride;
    const int64_t total = 2 * a.n_in;                     // bytes in this stream
    const int64_t ntiles = (a.n_out + a.tile - 1) / a.tile;
    const int64_t t_begin = (int64_t)blockIdx.x * a.tpw;
    const int64_t t_end = (t_begin + a.tpw < ntiles) ? t_begin + a.tpw : ntiles;

    // A workgroup walks a.tpw consecutive tiles of its stream (taps staged once). Requesting the next tile's
    // window into registers before filtering the current one was measured (with the clocks warm) and bought
    // nothing: the 6 workgroups per CU already overlap each other's staging and filtering.
    for (int64_t tile = t_begin; tile < t_end; tile++) {
        const TileGeom g = tile_geom(a, src, tile);
        stage_window(g, total, s_x, tid);
        __syncthreads();
        for (int k = tid; k < g.nouts; k += kThreads) {
            const uint8_t *x = s_x + g.head + 2 * (size_t)k * a.D;
            float acci = 0.f, accq = 0.f;
            if (a.arith && !(g.head & 1)) {
                if (a.mode == kDecimExact) tap_loop<kDecimExact>(a, x, s_taps, acci, accq);
                else if (a.mode == kDecimFma) tap_loop<kDecimFma>(a, x, s_taps, acci, accq);
                else tap_loop<kDecimFmaRaw>(a, x, s_taps, acci, accq);
            } else {
                for (int t = 0; t < a.L; t++) {
                    const float h = s_taps[t];
                    acci += s_lut[x[2 * t]] * h;
                    accq += s_lut[x[2 * t + 1]] * h;
                }
            }
            store_out(a, sid, g.k0 + k, acci, accq);
        }
        __syncthreads();
    }
}

// ---- the reference's two decimations (csdr fir_decimate_cc 45 / 50 at the default transition width: 79 taps) with every input
// sample CONVERTED ONCE per wave instead of once per output it feeds ---------------------------------------------------------------
// Output k reads samples [k D, k D + L): its last OV = L - D samples are output k + 1's first OV. In decim_kernel each lane converts all
// L samples of its output (2 cvt + 2 packed fma per sample: 4 of the 6 VALU instructions per tap). The scalar csdr loop sums in ascending
// tap order from zero, so the first OV terms of output k + 1 -- taps [0, OV) on exactly the samples output k's lane has just converted --
// are themselves a serial sum from zero: a PREFIX the neighbour can compute. Lane i of a wave owns output base + i - 1 and
//   1. converts its samples [D, L) into 2 OV registers;
//   2. forms with them the prefix of the NEXT output (taps [0, OV), ascending, from zero) and hands those two floats to lane i + 1
//      (one DPP move each);
//   3. continues its own output from the prefix it received: taps [OV, D) on samples nobody else needs (converted now), taps [D, L) on its
//      registers of step 1 -- one multiply and one add per tap and component, in the scalar loop's order.
// Lane 0 is the donor of lane 1's prefix (its own sum is discarded): a wave yields 63 outputs, a tile 4 x 63 = 252.
// Per output 2 OV x 2 + 2 OV + (D - OV) x 6 + 2 OV = 340 VALU instructions at D = 45 against 474, 45 LDS sample reads against 79, two
// independent sum chains per lane; the same float32 operations on the same operands in the same order: bit-identical outputs (f32 and
// s16; tested against decim_kernel and the oracle). PIRIP_DECIM_SHARED=0 (read at create) keeps decim_kernel for A/B runs.
typedef float dv2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float from_lane_down(float v)      // the value of lane - 1 (lane 0: 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
template <int D, int L>
__global__ __launch_bounds__(kThreads, 5) void decim_shared_kernel(DecimArgs a)
{
    constexpr int OV = L - D, kPerWave = 63, kWaves = kThreads / 64;
    static_assert(OV > 0 && OV < D && 2 * OV <= 96, "an output overlaps only its successor; the shared samples fit the register budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_taps = (float *)smem;                       // [L] (rounded up to 4)
    float *s_guard = s_taps + ((L + 3) & ~3);            // [256] floats nobody reads for a result: where the donor lane of a tile's first wave looks
    uint8_t *s_x = (uint8_t *)(s_guard + 256);           // staged u8 IQ, 16-B aligned
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sid = blockIdx.y;
    for (int i = tid; i < ((L + 3) & ~3); i += kThreads) s_taps[i] = i < L ? a.taps[i] : 0.0f;
    const uint8_t *src = a.in + (size_t)sid * a.in_stride;
    const int64_t total = 2 * a.n_in;
    const int64_t ntiles = (a.n_out + a.tile - 1) / a.tile;
    const int64_t t_begin = (int64_t)blockIdx.x * a.tpw;
    const int64_t t_end = (t_begin + a.tpw < ntiles) ? t_begin + a.tpw : ntiles;
    const dv2f chi = {a.c_hi, a.c_hi}, clo = {a.c_lo, a.c_lo}, m1 = {-1.0f, -1.0f};
    auto conv = [&](uint32_t w) {                        // csdr's x / 127.5 - 1 for the (I, Q) byte pair, as decim_kernel's tap_mac computes it
        const dv2f x = {(float)(w & 0xffu), (float)(w >> 8)};
        return __builtin_elementwise_fma(x, clo, __builtin_elementwise_fma(x, chi, m1));
    };
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    for (int64_t tile = t_begin; tile < t_end; tile++) {
        const TileGeom g = tile_geom(a, src, tile);
        stage_window(g, total, s_x, tid);
        __syncthreads();
        for (int base = wv * kPerWave; base < g.nouts; base += kWaves * kPerWave) {       // (wave-uniform)
            const int r = base + lane - 1;                                               // this lane's output in the tile; lane 0: the donor
            const uint8_t *x = s_x + g.head + 2 * r * D;
            // (the taps are read per chunk of outputs -- hoisted out of this loop they are 79 registers -- through an LDS address the
            //  compiler cannot see through, kept in the LDS address space: a laundered generic pointer turns the reads into flat loads)
            uint32_t taps = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
            asm volatile("" : "+v"(taps));
            auto tap4 = [&](int t0) { return *(const __attribute__((address_space(3))) f32x4_t *)(uintptr_t)(taps + 4u * (uint32_t)t0); };
            // steps 1 + 2: own samples [D, L), and with them the next output's prefix over taps [0, OV)
            dv2f sv[OV], pre = {0.f, 0.f};
#pragma unroll
            for (int u0 = 0; u0 < OV; u0 += 4) {
                const f32x4_t h4 = tap4(u0);
                const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int q = 0; q < 4 && u0 + q < OV; q++) {
                    const int u = u0 + q;
                    sv[u] = conv(lds_u16(x + 2 * (D + u)));
                    pre = pre + sv[u] * dv2f{hh[q], hh[q]};
                }
                // (pins the running sums in place: their only use is the store under the lane mask below, and the compiler otherwise sinks the
                //  whole accumulation under that branch -- behind every product, which it then has to spill)
                asm volatile("" : "+v"(pre));
            }
            // step 3: continue from the neighbour's prefix
            dv2f acc = {from_lane_down(pre.x), from_lane_down(pre.y)};
#pragma unroll
            for (int t0 = OV & ~3; t0 < L; t0 += 4) {
                const f32x4_t h4 = tap4(t0);
                const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int t = t0 + q;
                    if (t < OV || t >= L) continue;
                    const dv2f y = t < D ? conv(lds_u16(x + 2 * t)) : sv[t - D];
                    acc = acc + y * dv2f{hh[q], hh[q]};
                }
                asm volatile("" : "+v"(acc));
            }
            if (lane >= 1 && r < g.nouts) store_out(a, sid, g.k0 + r, acc.x, acc.y);
        }
        __syncthreads();
    }
}

// ---- small decimations (rtl_fsk's in-process decimator: 240 k / 40 k = 6, 1.8 M / 200 k = 9, / 180 k = 10, / 100 k = 18, 2.4 M / 80 k = 30:
// README.md:172,196,262,286, script/ping:47) -- the same idea carried through: with L = 79 taps an output spans S = ceil(L / D) blocks of
// D fresh samples, and every sample feeds S outputs. Lane l of a wave converts ONE block (block B0 + l) and the running sum of an output
// travels through the lanes like a systolic array: stage 0 starts output b from zero with taps [0, D) on its own block; after every stage
// the sums move one lane up (one DPP move per component) and stage s adds taps [s D, s D + D) on the block of the lane they arrived at. The
// sum that reaches lane l after stage S - 1 is output B0 + l - (S - 1), complete, its terms added in ascending tap order from zero -- the
// scalar csdr loop's order, so the float32 result is bit-identical. The first S - 1 lanes of a wave only feed their neighbours: a wave
// finishes 64 - (S - 1) outputs. Per output 4 D conversion + 2 L tap + 2 S move instructions against 6 L: D = 6: 210 / 0.80 (lane use) = 262
// against 474; D = 18: 240 / 0.94 = 255; D = 30: 282 / 0.97 = 291.
template <int D, int L>
__global__ __launch_bounds__(kThreads, 5) void decim_systolic_kernel(DecimArgs a)
{
    constexpr int S = (L + D - 1) / D, kPerWave = 64 - (S - 1), kWaves = kThreads / 64;
    static_assert(S >= 2 && kPerWave >= 32 && 2 * D <= 96, "several blocks per output, most lanes finish one, a block fits the registers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_taps = (float *)smem;                       // [L] (rounded up to 4)
    float *s_guard = s_taps + ((L + 3) & ~3);
    uint8_t *s_x = (uint8_t *)(s_guard + 256);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sid = blockIdx.y;
    for (int i = tid; i < ((L + 3) & ~3); i += kThreads) s_taps[i] = i < L ? a.taps[i] : 0.0f;
    const uint8_t *src = a.in + (size_t)sid * a.in_stride;
    const int64_t total = 2 * a.n_in;
    const int64_t ntiles = (a.n_out + a.tile - 1) / a.tile;
    const int64_t t_begin = (int64_t)blockIdx.x * a.tpw;
    const int64_t t_end = (t_begin + a.tpw < ntiles) ? t_begin + a.tpw : ntiles;
    const dv2f chi = {a.c_hi, a.c_hi}, clo = {a.c_lo, a.c_lo}, m1 = {-1.0f, -1.0f};
    auto conv = [&](uint32_t w) {
        const dv2f x = {(float)(w & 0xffu), (float)(w >> 8)};
        return __builtin_elementwise_fma(x, clo, __builtin_elementwise_fma(x, chi, m1));
    };
    for (int64_t tile = t_begin; tile < t_end; tile++) {
        const TileGeom g = tile_geom(a, src, tile);
        stage_window(g, total, s_x, tid);
        __syncthreads();
        for (int B0 = wv * kPerWave; B0 < g.nouts; B0 += kWaves * kPerWave) {             // (wave-uniform) first block of this wave's chunk
            const uint8_t *x = s_x + g.head + 2 * (B0 + lane) * D;                       // this lane's block of D fresh samples
            uint32_t taps = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
            asm volatile("" : "+v"(taps));
            dv2f y[D];
#pragma unroll
            for (int u = 0; u < D; u++) y[u] = conv(lds_u16(x + 2 * u));
            dv2f acc = {0.f, 0.f};
#pragma unroll
            for (int st = 0; st < S; st++) {
                if (st) acc = dv2f{from_lane_down(acc.x), from_lane_down(acc.y)};
#pragma unroll
                for (int u = 0; u < D; u++) {
                    const int t = st * D + u;
                    if (t >= L) break;
                    const float h = *(const __attribute__((address_space(3))) float *)(uintptr_t)(taps + 4u * (uint32_t)t);
                    acc = acc + y[u] * dv2f{h, h};
                }
                asm volatile("" : "+v"(acc));             // (pins the sum in place: see decim_shared_kernel)
            }
            const int o = B0 + lane - (S - 1);
            if (lane >= S - 1 && o < g.nouts) store_out(a, sid, g.k0 + o, acc.x, acc.y);
        }
        __syncthreads();
    }
}

// float-in variant for the libcsdr-compatible fir_decimate_cc(complexf*, ...) entry point
struct DecimFArgs { const float2 *in; float2 *out; const float *taps; int64_t n_out; int D, L; };
__global__ __launch_bounds__(kThreads) void decim_f32_kernel(DecimFArgs a)
{
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= a.n_out) return;
    const float2 *x = a.in + k * a.D;
    float acci = 0.f, accq = 0.f;
    for (int t = 0; t < a.L; t++) {
        const float h = a.taps[t];
        const float2 v = x[t];
        acci += v.x * h;
        accq += v.y * h;
    }
    a.out[k] = make_float2(acci, accq);
}

__global__ __launch_bounds__(kThreads) void cvt_u8_f_kernel(const uint8_t *in, float *out, const float *lut, int n)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) out[i] = s_lut[in[i]];
}

__global__ __launch_bounds__(kThreads) void cvt_f_s16_kernel(const float *in, short *out, int n)
{
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        out[i] = (short)(in[i] * (float)SHRT_MAX);
}

template <typename TI, typename TO, typename F>
void host_elementwise(const char *name, const TI *in, TO *out, int n, F launch)
{
    if (n <= 0) return;
    TI *d_in = nullptr; TO *d_out = nullptr;
    bool ok = hipMalloc((void **)&d_in, sizeof(TI) * (size_t)n) == hipSuccess &&
              hipMalloc((void **)&d_out, sizeof(TO) * (size_t)n) == hipSuccess &&
              hipMemcpy(d_in, in, sizeof(TI) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        launch(d_in, d_out);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpy(out, d_out, sizeof(TO) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (!ok) { fprintf(stderr, "pirip_hip %s: HIP failure (no usable GPU?) -- no CPU fallback\n", name); abort(); }
}

}  // namespace

struct pirip_hip_decim {
    int D = 0, L = 0, Lp = 0, out_s16 = 0, tile = 0, device = 0;
    size_t lds = 0;
    float c_hi = 0.f, c_lo = 0.f;   // exact arithmetic u8->float (see decim_kernel)
    int arith = 0;
    int mode = 0;                   // kDecimExact unless PIRIP_DECIM_FMA asked for a measurement variant at create
    int shared = 0;                 // 1: decim_shared_kernel<D, L>, 2: decim_systolic_kernel<D, L> exists for this shape (and PIRIP_DECIM_SHARED != 0): tile_sh / lds_sh are its geometry
    int tile_sh = 0; size_t lds_sh = 0;
    float tap_sum = 0.f;
    std::vector<float> taps;
    float *d_taps = nullptr, *d_lut = nullptr;
};

extern "C" {

int pirip_hip_decim_create(int decimation, float transition_bw, int out_s16, int device, pirip_hip_decim **out)
{
    if (!out || decimation < 1 || !(transition_bw > 0.f)) return PIRIP_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PIRIP_ERR_NO_DEVICE;
    if (device >= 0 && (device >= ndev || hipSetDevice(device) != hipSuccess)) return PIRIP_ERR_NO_DEVICE;
    pirip_hip_decim *d = new (std::nothrow) pirip_hip_decim();
    if (!d) return PIRIP_ERR_NOMEM;
    d->D = decimation; d->out_s16 = out_s16 ? 1 : 0;
    if (hipGetDevice(&d->device) != hipSuccess) { delete d; return PIRIP_ERR_NO_DEVICE; }
    d->L = csdr_filter_len(transition_bw);
    // csdr pads the taps with zeros to a multiple of 4 and uses the padded length in the
    // "enough input left" test; zero taps add +0 and are skipped in the kernel.
    d->Lp = d->L + 3 - ((d->L + 3) % 4);
    if (d->L > 4096) { delete d; return PIRIP_ERR_UNSUPPORTED; }
    d->taps.resize(d->L);
    csdr_lowpass_hamming(d->taps.data(), d->L, 0.5 / (float)decimation);
    // tile: as many outputs per workgroup as fit a 48 KiB u8 window
    int tile = kThreads;
    while (tile > 1 && 2 * ((size_t)(tile - 1) * d->D + d->L) + 32 > 48 * 1024) tile /= 2;
    d->tile = tile;
    d->lds = sizeof(float) * (((size_t)d->L + 3) & ~(size_t)3) + sizeof(float) * 256 +
             ((2 * ((size_t)(tile - 1) * d->D + d->L) + 47) & ~(size_t)15);
    std::vector<float> lut(256);
    for (int x = 0; x < 256; x++) lut[x] = ((float)x) / (UCHAR_MAX / 2.0) - 1.0;   // convert_u8_f
    bool ok = hipMalloc((void **)&d->d_taps, sizeof(float) * d->L) == hipSuccess &&
              hipMalloc((void **)&d->d_lut, sizeof(float) * 256) == hipSuccess &&
              hipMemcpy(d->d_taps, d->taps.data(), sizeof(float) * d->L, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d->d_lut, lut.data(), sizeof(float) * 256, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { if (d->d_taps) (void)hipFree(d->d_taps); if (d->d_lut) (void)hipFree(d->d_lut); delete d; return PIRIP_ERR_NOMEM; }
    // arithmetic u8->float must reproduce csdr's double formula for every byte value, else keep the table
    {
        d->c_hi = (float)(std::nearbyint((1.0 / 127.5) * 4194304.0) / 4194304.0);
        d->c_lo = (float)(1.0 / 127.5 - (double)d->c_hi);
        bool exact = true;
        for (int x = 0; x < 256; x++) {
            const float y = std::fmaf((float)x, d->c_lo, std::fmaf((float)x, d->c_hi, -1.0f));
            exact &= (y == lut[x]);
        }
        d->arith = exact && !getenv("PIRIP_DECIM_LUT");
        if (const char *e = getenv("PIRIP_DECIM_FMA")) {
            const int m = atoi(e);
            if (d->arith && (m == kDecimFma || m == kDecimFmaRaw)) d->mode = m;
        }
        double hs = 0.0;
        for (float h : d->taps) hs += (double)h;
        d->tap_sum = (float)hs;
    }
    {
        const char *e = getenv("PIRIP_DECIM_SHARED");
        const bool on = (!e || atoi(e)) && d->arith && d->L == 79;
        d->shared = on && (d->D == 45 || d->D == 50) ? 1 : on && (d->D == 6 || d->D == 9 || d->D == 10 || d->D == 18 || d->D == 30) ? 2 : 0;   // 1: decim_shared_kernel, 2: decim_systolic_kernel
        d->tile_sh = d->shared == 2 ? 4 * (64 - ((d->L + d->D - 1) / d->D - 1)) : 4 * 63;
        d->lds_sh = sizeof(float) * (((size_t)d->L + 3) & ~(size_t)3) + sizeof(float) * 256 + ((2 * ((size_t)(d->tile_sh - 1) * d->D + d->L) + 47) & ~(size_t)15);
    }
    if (d->lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)decim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->lds) != hipSuccess) {
        (void)hipFree(d->d_taps); (void)hipFree(d->d_lut); delete d; return PIRIP_ERR_HIP;
    }
    *out = d;
    return PIRIP_OK;
}

int pirip_hip_decim_destroy(pirip_hip_decim *d)
{
    if (!d) return PIRIP_ERR_BAD_ARG;
    (void)hipDeviceSynchronize();
    (void)hipFree(d->d_taps); (void)hipFree(d->d_lut);
    delete d;
    return PIRIP_OK;
}

int pirip_hip_decim_set_arith(pirip_hip_decim *d, int mode)
{
    if (!d || mode < kDecimExact || mode > kDecimFmaRaw) return PIRIP_ERR_BAD_ARG;
    if (mode != kDecimExact && !d->arith) return PIRIP_ERR_UNSUPPORTED;     // the table path has one arithmetic
    d->mode = mode;
    return PIRIP_OK;
}
int pirip_hip_decim_get_arith(const pirip_hip_decim *d) { return d ? d->mode : PIRIP_ERR_BAD_ARG; }

int pirip_hip_decim_taps(const pirip_hip_decim *d, float *taps, int *ntaps)
{
    if (!d || !ntaps) return PIRIP_ERR_BAD_ARG;
    if (taps) std::memcpy(taps, d->taps.data(), sizeof(float) * d->L);
    *ntaps = d->L;
    return PIRIP_OK;
}

int64_t pirip_hip_decim_nout(const pirip_hip_decim *d, int64_t n_in)
{
    if (!d || n_in < d->Lp) return 0;
    return (n_in - d->Lp) / d->D + 1;
}

int pirip_hip_decim_batch(pirip_hip_decim *d, const uint8_t *d_in, size_t in_stride_bytes, int64_t n_in,
                          void *d_out, size_t out_stride_bytes, int nstreams, void *hip_stream)
{
    if (!d || !d_in || !d_out || nstreams <= 0 || n_in < 0) return PIRIP_ERR_BAD_ARG;
    const int64_t n_out = pirip_hip_decim_nout(d, n_in);
    if (n_out <= 0) return PIRIP_OK;
    int cur = -1;   // run on the device the stage was created on
    if ((hipGetDevice(&cur) != hipSuccess || cur != d->device) && hipSetDevice(d->device) != hipSuccess) return PIRIP_ERR_NO_DEVICE;
    // the shape-specialised kernel (every sample converted once per wave) where it exists: exact arithmetic, 16-bit aligned windows
    const bool sh = d->shared && d->mode == kDecimExact && !(((uintptr_t)d_in | (uintptr_t)in_stride_bytes) & 1);
    const int tile = sh ? d->tile_sh : d->tile;
    DecimArgs a{d_in, in_stride_bytes, n_in, d_out, out_stride_bytes, n_out, d->d_taps, d->d_lut, d->D, d->L, tile, d->out_s16, d->arith, 0, d->mode, d->c_hi, d->c_lo, d->tap_sum};
    const int64_t ntiles = (n_out + tile - 1) / tile;
    // tiles per workgroup: a long walk (read-ahead, taps staged once) as long as the chip stays many times over-filled
    int tpw = 8;
    if (const char *e = getenv("PIRIP_DECIM_TPW")) tpw = atoi(e) > 0 ? atoi(e) : tpw;
    while (tpw > 1 && ((ntiles + tpw - 1) / tpw) * nstreams < 8 * 256 * 6) tpw /= 2;
    a.tpw = tpw;
    const int64_t nwg = (ntiles + tpw - 1) / tpw;
    if (nwg > 0x7fffffff) return PIRIP_ERR_UNSUPPORTED;
#define PIRIP_SYS_LAUNCH(DD) hipLaunchKernelGGL((decim_systolic_kernel<DD, 79>), dim3((unsigned)nwg, (unsigned)nstreams), dim3(kThreads), d->lds_sh, (hipStream_t)hip_stream, a)
    if (sh && d->shared == 2) {
        switch (d->D) { case 6: PIRIP_SYS_LAUNCH(6); break; case 9: PIRIP_SYS_LAUNCH(9); break; case 10: PIRIP_SYS_LAUNCH(10); break;
                        case 18: PIRIP_SYS_LAUNCH(18); break; default: PIRIP_SYS_LAUNCH(30); break; }
    }
#undef PIRIP_SYS_LAUNCH
    else if (sh && d->D == 45) hipLaunchKernelGGL((decim_shared_kernel<45, 79>), dim3((unsigned)nwg, (unsigned)nstreams), dim3(kThreads), d->lds_sh, (hipStream_t)hip_stream, a);
    else if (sh) hipLaunchKernelGGL((decim_shared_kernel<50, 79>), dim3((unsigned)nwg, (unsigned)nstreams), dim3(kThreads), d->lds_sh, (hipStream_t)hip_stream, a);
    else hipLaunchKernelGGL(decim_kernel, dim3((unsigned)nwg, (unsigned)nstreams), dim3(kThreads), d->lds,
                            (hipStream_t)hip_stream, a);
    return hipGetLastError() == hipSuccess ? PIRIP_OK : PIRIP_ERR_HIP;
}

// ---- section D: libcsdr-compatible host-buffer entry points ------------------------------------
void convert_u8_f(unsigned char *input, float *output, int length)
{
    // one table per process, built on first use (function-local static: initialised once, thread-safe)
    static float *const d_lut = [] {
        std::vector<float> lut(256);
        for (int x = 0; x < 256; x++) lut[x] = ((float)x) / (UCHAR_MAX / 2.0) - 1.0;
        float *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(float) * 256) != hipSuccess ||
            hipMemcpy(p, lut.data(), sizeof(float) * 256, hipMemcpyHostToDevice) != hipSuccess) {
            fprintf(stderr, "pirip_hip convert_u8_f: no usable HIP device -- no CPU fallback\n"); abort();
        }
        return p;
    }();
    const float *lutp = d_lut;
    host_elementwise("convert_u8_f", (const uint8_t *)input, output, length, [&](uint8_t *di, float *dout) {
        int blocks = (length + kThreads - 1) / kThreads; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(cvt_u8_f_kernel, dim3(blocks), dim3(kThreads), 0, nullptr, di, dout, lutp, length);
    });
}

void convert_f_s16(float *input, short *output, int length)
{
    host_elementwise("convert_f_s16", (const float *)input, output, length, [&](float *di, short *dout) {
        int blocks = (length + kThreads - 1) / kThreads; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(cvt_f_s16_kernel, dim3(blocks), dim3(kThreads), 0, nullptr, di, dout, length);
    });
}

int firdes_filter_len(float transition_bw) { return csdr_filter_len(transition_bw); }
void firdes_lowpass_f_hamming(float *output, int length, float cutoff_rate) { csdr_lowpass_hamming(output, length, cutoff_rate); }
void firdes_lowpass_f(float *output, int length, float cutoff_rate, window_t window) { csdr_lowpass(output, length, cutoff_rate, (int)window); }

int fir_decimate_cc(complexf *input, complexf *output, int input_size, int decimation, float *taps, int taps_length)
{
    if (!input || !output || !taps || input_size < taps_length || decimation < 1 || taps_length < 1) return 0;
    const int64_t n_out = ((int64_t)input_size - taps_length) / decimation + 1;
    float2 *d_in = nullptr, *d_out = nullptr; float *d_taps = nullptr;
    bool ok = hipMalloc((void **)&d_in, sizeof(float2) * (size_t)input_size) == hipSuccess &&
              hipMalloc((void **)&d_out, sizeof(float2) * (size_t)n_out) == hipSuccess &&
              hipMalloc((void **)&d_taps, sizeof(float) * (size_t)taps_length) == hipSuccess &&
              hipMemcpy(d_in, input, sizeof(float2) * (size_t)input_size, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_taps, taps, sizeof(float) * (size_t)taps_length, hipMemcpyHostToDevice) == hipSuccess;
    int ret = 0;
    if (ok) {
        DecimFArgs a{d_in, d_out, d_taps, n_out, decimation, taps_length};
        hipLaunchKernelGGL(decim_f32_kernel, dim3((unsigned)((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, nullptr, a);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpy(output, d_out, sizeof(float2) * (size_t)n_out, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) ret = (int)n_out;
    }
    if (!ok) fprintf(stderr, "pirip_hip fir_decimate_cc: HIP failure (no usable GPU?) -- no CPU fallback\n");
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_taps) (void)hipFree(d_taps);
    return ret;
}

}  // extern "C"
