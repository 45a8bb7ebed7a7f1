// pirip_amd/csrc/fsk_demod_wave.hip -- wave-per-stream FSK demodulator for gfx950 (MI355X), second generation.
//
// One wavefront owns one IQ stream and walks its frames in order (the demodulator is frame-serial: nin, the smoothed
// spectrum Sf, the tone estimates, the oscillator phases and the integrator memory chain frame to frame,
// [UPSTREAM-RECALLED codec2 fsk.c: fsk_demod_freq_est + fsk_demod_core; SURVEY.md 8a rows a-1, a-5 ... a-8]).
// Instances (table kInst at the end of the file, keyed by samples per symbol) cover the reference's receive command lines:
//   Ts = 24, Ndft = 256 : fsk_demod -d -p 24 2 240000 10000 (README.md:105, test/loopback_rtl_sdr.sh:16), default P = 8,
//                          rtl_fsk's P = 6 with csdr's u8 conversion (test/loopback_rtl_fsk.sh:10), 2- and 4-FSK (config 4)
//   Ts = 40, Ndft = 512 : fsk_demod -c 2 40000 1000 behind csdr fir_decimate_cc 45 | convert_f_s16 (README.md:109) and the
//                          services' modem rtl_fsk -a 40000 -r 1000 (script/ping:47, script/frame_repeater:36), P = 8 / 10,
//                          2- and 4-FSK (README.md:232-239)
//   Ts = 20 / 18, Ndft = 256 : rtl_fsk -a 200000 / -a 180000 -r 10000 (README.md:262,286,292,297), float samples, 2- and 4-FSK
//   MASK                 : the `--mask` comb estimator on every shape but P = 24 (README.md:239-297)
//
// What the measurements (tools/valu_issue_bench.hip -> profiles/r02_valu_issue.txt, profiles/r02_power_trace.txt) say about
// this machine: ONE wave issues at most one VALU instruction per ~5 cycles (8.5 when it depends on the previous one), a
// SIMD sustains one per ~1.7 (plain f32) / ~2.8 (packed f32, DPP) shader cycles with 4 waves -- but the shader clock drops
// under that load (1.3-1.9 GHz), and in WALL time a SIMD retires a plain op per ~2.9, a packed / DPP / conversion op per
// ~4.5-5 and a transcendental per ~8.3 "2.4 GHz cycles" whether 2, 3 or 4 waves are resident. So: enough waves to cover
// LDS latency (3 per SIMD on the headline shape), and then as few executed instructions as the exact arithmetic allows
// (hand-packed f32 pairs where hipcc's vectoriser shuffles, DPP reductions, a 6-instruction correctly rounded sqrt).
// The kernel is laid out to need few registers and little LDS per wave:
//   * a workgroup is WPB waves = WPB streams that share the read-only FFT tables in LDS (one barrier after the table
//     load, none afterwards: streams never wait for each other);
//   * the frame's raw samples are staged linearly in LDS by LDS-DMA (buffer_load_dwordx4 ... lds, bounds-checked by the
//     buffer descriptor): no staging registers, every input byte read from HBM once, the next frame's superset
//     (N + Ts/4 samples from the known frame start) requested as soon as the correlator has read this frame's and
//     landing while window sums, timing and decisions run;
//   * "old" integrator-memory positions (last frame's samples) sit in a guard in front of the staged frame: the RAW samples
//     of the last frame's tail, copied there before the next frame's DMA overwrites them, and mixed again with last frame's
//     tone estimates continued backwards from the phase reference (round 4: no f_dc ring in LDS, no per-sample LDS traffic
//     in the correlator, 1.3 KB (2-FSK) / 2.7 KB (4-FSK) less LDS per stream -- the 4-FSK instances now fit three blocks per CU);
//   * all window prefix sums live in registers; the two the decision stage needs are picked by a uniform branch tree
//     instead of a 24-way v_cndmask select.
// Numerics as in DESIGN.md: Sf, f_est, nin bit-exact (kiss_fft dataflow, no fused multiply-add on that path: file
// built with -ffp-contract=off), f_dc / f_int / rx_filt within tolerance (per-lane restart of the upstream oscillator
// recursion with its first-order gain drift).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../include/pirip_hip.h"
#include "fsk_device.hpp"

namespace pirip {

namespace {

constexpr int kWave = 64;

struct cf { float x, y; };
typedef float v2f __attribute__((ext_vector_type(2)));

// Ordering point for LDS traffic inside ONE wavefront: LDS instructions of a wave execute in issue order, so only the
// compiler has to be kept from reordering (see fsk_demod_general.hip for why __syncthreads() is the wrong tool here).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// b + (a of lane + 1): one VALU instruction with a DPP source (hipcc does not fold wave_shl moves into the consumer)
__device__ __forceinline__ float add_lane_up(float a, float b)
{
    float r;
    asm("v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#define PIRIP_DPP_F(old, src, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), ctrl, rmask, 0xf, false))
#define PIRIP_DPP_I(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, rmask, 0xf, false)
// value of lane + 1 (lane 63 reads 0: bound_ctrl, so the instruction has no tied "old" operand and hipcc can fold it
// into the consuming VALU op as a DPP source instead of v_mov + v_mov_dpp)
__device__ __forceinline__ float lane_up(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ float wsum(float v)
{
    v += PIRIP_DPP_F(0.f, v, 0x111, 0xf);   // row_shr:1
    v += PIRIP_DPP_F(0.f, v, 0x112, 0xf);   // row_shr:2
    v += PIRIP_DPP_F(0.f, v, 0x114, 0xf);   // row_shr:4
    v += PIRIP_DPP_F(0.f, v, 0x118, 0xf);   // row_shr:8
    v += PIRIP_DPP_F(0.f, v, 0x142, 0xa);   // row_bcast:15
    v += PIRIP_DPP_F(0.f, v, 0x143, 0xc);   // row_bcast:31 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// arg-max with codec2's tie rule (first maximum wins): larger value, then smaller index. v >= 0 and never NaN (a lane's
// candidate is only ever replaced by "w > best" with best starting at 0), so the wave maximum is six v_max_f32 with a DPP
// source and the winning index is the minimum index among the lanes that hold the maximum (six v_min_i32). A lane whose
// DPP source is outside its row, or whose row is masked out, is not written and keeps its own value; lane 63 ends up with
// the reduction over the wave. s_nop 1: VALU write -> DPP read needs two wait states and hipcc does not look inside asm.
#define PIRIP_DPP_REDUCE(op) \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        "s_nop 1\n\tv_readlane_b32 %1, %0, 63"
__device__ __forceinline__ void wargmax(float &v, int &idx)
{
    float red = v;
    int smax, smin;
    asm(PIRIP_DPP_REDUCE("v_max_f32_dpp") : "+v"(red), "=s"(smax));
    int cand = (__builtin_bit_cast(int, v) == smax) ? idx : 0x7fffffff;      // v >= 0: equal values <=> equal bit patterns
    asm(PIRIP_DPP_REDUCE("v_min_i32_dpp") : "+v"(cand), "=s"(smin));
    v = __builtin_bit_cast(float, smax);
    idx = smin;
}
#undef PIRIP_DPP_REDUCE

__device__ __forceinline__ float ubyte0(uint32_t v) { return (float)(v & 0xffu); }
__device__ __forceinline__ float ubyte1(uint32_t v) { return (float)((v >> 8) & 0xffu); }
__device__ __forceinline__ float ubyte2(uint32_t v) { return (float)((v >> 16) & 0xffu); }
__device__ __forceinline__ float ubyte3(uint32_t v) { return (float)(v >> 24); }

// Correctly rounded sqrt for x that is zero or >= 2^-96; the caller takes this path only when every value of the batch
// qualifies (one wave-uniform test), sqrtf() otherwise.
//   FINITE: q = min(rsq(x), 2^60); y = x q; result = fma(fma(-y, y, x), q/2, y) -- one transcendental + 5 VALU. Correct
//           rounding is not a theorem but a measurement: tools/compiler_checks.hip and the library's self-test
//           (pirip_hip_selftest_sqrt, run by the GPU tests) compare it with (float)sqrt((double)x) for x = 0 and EVERY float in
//           [2^-96, FLT_MAX] on the device; the clamp makes x = 0 give 0 and is a no-op elsewhere. +inf would give NaN,
//           so the f32 input format (the only one that can produce an infinite |X|^2) keeps
//   general: v_sqrt_f32 (within 1 ulp) plus the neighbour-residual test -- one transcendental + 8 VALU, inf/NaN as sqrtf.
//   NZ (every value of the batch >= 2^-96, none zero: what a live receiver sees): the FINITE form without the clamp, which is a no-op
//           there (rsq(2^-96) = 2^48) -- 4 instead of 5 VALU; and the batch's range test is then the minimum of the raw bit patterns
//           (non-negative floats order as unsigned integers), without the "- 1" per value that lets zero pass (sqrt_key).
template <bool FINITE, bool NZ = false>
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    if (FINITE) {
        float q = __builtin_amdgcn_rsqf(x);
        if (!NZ) asm("v_min_f32 %0, %0, %1" : "+v"(q) : "v"(0x1p60f));
        const float y = x * q, h = 0.5f * q;
        return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
    }
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float rm = __builtin_fmaf(-ym, y, x);
    const float rp = __builtin_fmaf(-yp, y, x);
    float r = (rm <= 0.0f) ? ym : y;
    r = (rp > 0.0f) ? yp : r;
    return r;
}
// llr_frame_gain (fsk_device.hpp) with its square root in the v_sqrt + neighbour-residual form where the argument allows it (wave-uniform)
__device__ __forceinline__ float llr_frame_gain_quick(int llr_map, float sig, float nse)
{
    const float a2 = sig - nse;
    if (!(a2 >= 0x1p-96f && a2 <= 0x1p126f)) return llr_frame_gain(llr_map, sig, nse);
    const float amp = sqrt_rn_normal<false>(a2);
    if (llr_map == kLlrRician) return (2.0f * amp) / nse;
    return (2.0f * (sig / nse)) / amp;
}
__device__ __forceinline__ unsigned sqrt_key(float x) { return __builtin_bit_cast(unsigned, x) - 1u; }
__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
constexpr unsigned kSqrtLo = 0x0f800000u;                           // 2^-96
__device__ __forceinline__ unsigned umin2(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---- packed-f32 complex helpers ------------------------------------------------------------------------------------
// A complex value is one VGPR pair (x = re in the low half). gfx950's v_pk_*_f32 take per-operand half selectors
// (op_sel / op_sel_hi) and per-half negation (neg_lo / neg_hi), so kiss_fft's complex multiply is 3 instructions and the
// +-j rotation inside the radix-4 butterfly is free; hipcc builds those operand swizzles with v_mov/v_xor copies, hence the
// inline asm (one asm statement per helper: hipcc pads adjacent asm statements that feed each other with s_nop).
// kiss_fft C_MUL: (a.x*t.x - a.y*t.y, a.x*t.y + a.y*t.x): three instructions, each product/sum rounded once (no fma)
__device__ __forceinline__ v2f cmul_x(v2f a, v2f t)
{
    v2f p2, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]"
        : "=&v"(r), "=&v"(p2) : "v"(a), "v"(t));
    return r;
}
// real window sample (low / high half of a table pair) times a complex sample: one packed multiply with a broadcast selector
// (written as scalar code hipcc moves the odd-numbered window samples into low halves first)
__device__ __forceinline__ v2f scale_lo(v2f h, v2f x)
{
    v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(h), "v"(x)); return r;
}
__device__ __forceinline__ v2f scale_hi(v2f h, v2f x)
{
    v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(h), "v"(x)); return r;
}
// |w|^2 = (w.x*w.x) + (w.y*w.y), each product and the sum rounded once: one packed multiply and one add of its halves
// (left to the SLP vectoriser, pairs of these become 3 v_mov shuffles + 3 packed ops)
__device__ __forceinline__ float mag2(v2f w)
{
    v2f sq;
    asm("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(w));
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(sq.x), "v"(sq.y));
    return r;
}
// two spectrum-smoothing updates, Sf = (Sf * (1 - tc)) + (|X| * tc) with k = (1 - tc, tc): three packed ops, every product and
// the sum rounded once (hipcc's own vectorisation of the scalar form pairs Sf with |X| and pays three v_mov per update)
__device__ __forceinline__ v2f smooth2(v2f sf, v2f mag, v2f k)
{
    v2f t;
    asm("v_pk_mul_f32 %0, %0, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(sf), "=&v"(t) : "v"(mag), "v"(k));
    return sf;
}
__device__ __forceinline__ v2f add_rot(v2f a, v2f b)   // a + (b.y, -b.x)
{
    v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ v2f sub_rot(v2f a, v2f b)   // a - (b.y, -b.x)
{
    v2f r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// down-conversion x * conj(ph): 2 packed ops (fma allowed here)
__device__ __forceinline__ v2f mix_conj(v2f x, v2f ph)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(x), "v"(ph));
    return r;
}
// acc + x * conj(ph): the down-conversion folded into the running sum, 2 packed fma (round 5: was mix_conj + one packed add)
__device__ __forceinline__ v2f mix_conj_acc(v2f x, v2f ph, v2f acc)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(x), "v"(ph), "v"(acc));
    return r;
}
// oscillator step ph * d: 2 packed ops
__device__ __forceinline__ v2f rot_step(v2f ph, v2f d)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(ph), "v"(d));
    return r;
}
// kiss_fft radix-4 butterfly (forward) on operands already multiplied by their twiddles
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

// ---- input formats ------------------------------------------------------------------------------------------------
// Exact conversions (each checked against the defining expression for every input value in tests/test_boundary_cpu.py):
//   fsk_demod -d   (x - 127)/128              = fma(x, 2^-7, -127/128)
//   csdr / rtl_fsk x/127.5 - 1 (double, rounded) = fma(x, c_lo, fma(x, c_hi, -1)), c_hi a multiple of 2^-22
//   fsk_demod -c   x/750                       = fma(x, c_lo, x*c_hi), c_hi = 175/2^17 (x*c_hi exact for every int16)
template <int FMT> struct InFmt;
template <> struct InFmt<PIRIP_IN_CU8_FSKDEMOD> { static constexpr int BPS = 2; static constexpr bool NEUTRAL_OK = true; static constexpr uint32_t NEUTRAL = 0x7F7F7F7Fu; };
template <> struct InFmt<PIRIP_IN_CU8_CSDR> { static constexpr int BPS = 2; static constexpr bool NEUTRAL_OK = false; static constexpr uint32_t NEUTRAL = 0x80808080u; };
template <> struct InFmt<PIRIP_IN_CS16> { static constexpr int BPS = 4; static constexpr bool NEUTRAL_OK = true; static constexpr uint32_t NEUTRAL = 0u; };
template <> struct InFmt<PIRIP_IN_CF32> { static constexpr int BPS = 8; static constexpr bool NEUTRAL_OK = true; static constexpr uint32_t NEUTRAL = 0u; };

template <int FMT>
__device__ __forceinline__ float cvt_u8(float b)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD) return __builtin_fmaf(b, 0.0078125f, -0.9921875f);
    return __builtin_fmaf(b, -1.187418e-07f, __builtin_fmaf(b, 0.007843255996704102f, -1.0f));
}
// the same maps on an (I, Q) pair of byte values: one v_pk_fma_f32 per fma instead of two scalar ones (round 5)
template <int FMT>
__device__ __forceinline__ v2f cvt_u8_pair(v2f b)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD) return __builtin_elementwise_fma(b, v2f{0.0078125f, 0.0078125f}, v2f{-0.9921875f, -0.9921875f});
    return __builtin_elementwise_fma(b, v2f{-1.187418e-07f, -1.187418e-07f},
                                     __builtin_elementwise_fma(b, v2f{0.007843255996704102f, 0.007843255996704102f}, v2f{-1.0f, -1.0f}));
}
__device__ __forceinline__ float cvt_s16(float x)
{
    const float hi = x * 0.00133514404296875f;                    // exact (16-bit x 8-bit significands)
    return __builtin_fmaf(x, -0x1.e60f04p-20f, hi);
}
// sample k of a block held as dwords in rw[] (k is a compile-time constant after unrolling)
template <int FMT>
__device__ __forceinline__ v2f decode(const uint32_t *rw, int k)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD || FMT == PIRIP_IN_CU8_CSDR) {
        const uint32_t v = rw[k >> 1];
        return cvt_u8_pair<FMT>(v2f{(k & 1) ? ubyte2(v) : ubyte0(v), (k & 1) ? ubyte3(v) : ubyte1(v)});
    } else if (FMT == PIRIP_IN_CS16) {
        const uint32_t v = rw[k];
        return v2f{cvt_s16((float)(short)(v & 0xffffu)), cvt_s16((float)((int)v >> 16))};
    } else {
        return v2f{__builtin_bit_cast(float, rw[2 * k]), __builtin_bit_cast(float, rw[2 * k + 1])};
    }
}
// one sample at an LDS byte address (FFT input gather)
template <int FMT>
__device__ __forceinline__ v2f lds_sample(const unsigned char *p)
{
    if (FMT == PIRIP_IN_CU8_FSKDEMOD || FMT == PIRIP_IN_CU8_CSDR) {
        const uint32_t v = *(const uint16_t *)p;
        return cvt_u8_pair<FMT>(v2f{ubyte0(v), ubyte1(v)});
    } else if (FMT == PIRIP_IN_CS16) {
        const uint32_t v = *(const uint32_t *)p;
        return v2f{cvt_s16((float)(short)(v & 0xffffu)), cvt_s16((float)((int)v >> 16))};
    } else {
        const float2 v = *(const float2 *)p;
        return v2f{v.x, v.y};
    }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// atan2(y, x) / (2 pi) in turns, finite arguments (the bounded input formats: 8-bit and s16 samples cannot overflow the timing sum):
// t = min / max by v_rcp, atan(t) / 2 pi = t q(t^2) with a degree-7 minimax q evaluated Estrin-fashion (dependent depth 5 instead of
// the ~40 of ocml's atan2f with its IEEE divide -- this block runs once per frame on the wave's critical path), then the octant /
// quadrant / sign reflections. Largest error against the exact value 4.5e-8 turns over 2 * 10^6 random arguments (float
// atan2f / 2 pi: 6.6e-8) -- a fine-timing estimate is compared at 5e-5.
__device__ __forceinline__ float atan2_turns(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mn = ax < ay ? ax : ay, mx = ax < ay ? ay : ax;
    const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float s1 = t * t, s2 = s1 * s1, s4 = s2 * s2;
    const float p01 = __builtin_fmaf(s1, -0.05304612219333649f, 0.15915483236312866f);
    const float p23 = __builtin_fmaf(s1, -0.02213626727461815f, 0.031745944172143936f);
    const float p45 = __builtin_fmaf(s1, -0.00889870710670948f, 0.015346021391451359f);
    const float p67 = __builtin_fmaf(s1, -0.0006453014793805778f, 0.0034795869141817093f);
    const float q0 = __builtin_fmaf(s2, p23, p01), q1 = __builtin_fmaf(s2, p67, p45);
    float r = __builtin_fmaf(s4, q1, q0) * t;
    r = ay > ax ? 0.25f - r : r;
    r = x < 0.f ? 0.5f - r : r;
    return __builtin_copysignf(r, y);
}

// ---- fused FSK_LDPC hand-over (SoftOut): the arithmetic of ldpc_kernels.hip's LLR stage / the checker (ldpc_oracle.c), operation for operation
constexpr float kLlrMax = 24.0f;
// ln I0(x), x >= 0: table at multiples of 1/8 up to 32 with linear interpolation, slope 1 beyond
// (branch-free, so that a lane's four look-ups are in flight together: beyond 32 the argument is held at 32, where the
//  interpolation gives tab[256] + 0 exactly, and x - 32 is added -- the value the two-branch form returns)
__device__ __forceinline__ float ln_i0_tab(const float *tab, float x)
{
    const bool in = x < 32.0f;
    const float xs = (in ? x : 32.0f) * 8.0f;
    const int j = (int)xs;
    const float f = xs - (float)j;
    const float t0 = tab[j], t1 = tab[j + 1];
    return (t0 + (f * (t1 - t0))) + (in ? 0.0f : x - 32.0f);
}
__device__ __forceinline__ uint32_t spread16(uint32_t x)     // bit t of x -> bit 2 t
{
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// Per-phase cycle split (profiling builds only: -DPIRIP_WAVE_TIMING): s_memtime deltas of stream 0's wave are summed per phase
// and written over the first frames' stats rows at the end (tools/phase_split.py reads them). Costs ~10 % by itself.
// PIRIP_WAVE_STOP=k in the environment of such a build ends every frame at mark k (1: after the estimator FFTs, 2: peak pick,
// 3: correlator, 4: window sums + timing; anything else: whole frame) behind a run-time test the compiler cannot see through, so the
// phases before the mark are compiled exactly as in the full frame: differences of SQ_INSTS_VALU between consecutive k are the
// instructions each phase EXECUTES (tools/phase_valu.sh). Results of a stopped run are wrong by construction.
#ifdef PIRIP_WAVE_TIMING
__device__ int g_wave_stop_phase = -1;
#define PIRIP_T_DECL long long t_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long t_last_ = __builtin_amdgcn_s_memtime(); \
    const int t_stop_ = __builtin_amdgcn_readfirstlane(g_wave_stop_phase)
#define PIRIP_T_MARK(i) { const long long t_now_ = __builtin_amdgcn_s_memtime(); t_acc_[i] += t_now_ - t_last_; t_last_ = t_now_; } \
    if (t_stop_ == (i)) { pos += nin; frame++; wave_lds_sync(); continue; } else (void)0
#else
#define PIRIP_T_DECL do { } while (0)
#define PIRIP_T_MARK(i) do { } while (0)
#endif


#ifndef PIRIP_XPS256            // (build-time experiment knobs: bytes per 16x16 transpose group, waves per SIMD of the Ts = 24 4-FSK instances)
#define PIRIP_XPS256 2176
#endif
#ifndef PIRIP_M4_WPS
#define PIRIP_M4_WPS 3
#endif
#ifndef PIRIP_S16_DIRECT        // the same for the complex-s16 instances (Ts = 40: 8.5 KB of staging per stream, two blocks per CU): measured, interleaved
#define PIRIP_S16_DIRECT 0      // A/B: 254 against 261 G samples/s (2-FSK), 196 against 209 G (4-FSK mask) -- staged stays
#endif
#ifndef PIRIP_F32_PREFETCH
#define PIRIP_F32_PREFETCH 1
#endif
#ifndef PIRIP_F32_DIRECT        // 1: complex-float instances read their samples from global memory instead of staging the frame in LDS
#define PIRIP_F32_DIRECT 1
#endif
template <int M, int TS, int P, int NSYM, int NDFT, int FMT>
struct WaveCfg {
    static constexpr int BPS = InFmt<FMT>::BPS;
    static constexpr int N = TS * NSYM;
    static constexpr int NMEM = N + 2 * TS;
    static constexpr int Q = TS / 4;
    static constexpr int HIST = 2 * TS + Q;
    static constexpr int STEP = TS / P;
    static constexpr int NLANES = NSYM + 2;                    // symbol blocks per frame, one lane each
    static constexpr int NFFT = (N - Q) / (NDFT / 2) - 1;
    // raw staging: [guard of neutral samples: HIST of them, 16-byte granules][superset of N + Q samples][DMA slack]
    static constexpr int GUARD_B = ((HIST * BPS + 15) / 16) * 16;
    static constexpr int SUP_B = (N + Q) * BPS;
    static constexpr int NDMA16 = SUP_B / 1024;                 // 64 lanes x 16 bytes per instruction
    static constexpr int NDMA4 = (SUP_B - NDMA16 * 1024 + 255) / 256;
    // complex-float input (8 bytes per sample) is NOT staged: a frame would be 16 KB of LDS per stream at Ts = 40 and hold the f32 instances
    // at 6 waves per CU. Its FFT inputs and the correlator blocks come straight from global memory (L2); LDS keeps the guard and, behind
    // it, the HEAD: the first Ts + Ts/4 new samples, so that the three blocks that contain last frame's positions (lanes 0..2) read one
    // linear piece of LDS exactly as in the staged layout. DUMP: where the line-touching prefetch of the next frame lands.
    // (Ndft = 128 shapes -- Ts = 8, 10: 4 KB frames -- stay staged: measured 3-10 % slower unstaged)
    static constexpr bool DIRECT = (PIRIP_F32_DIRECT != 0) && (FMT == PIRIP_IN_CF32 || (FMT == PIRIP_IN_CS16 && PIRIP_S16_DIRECT != 0)) && NDFT >= 256;
    static constexpr int HEAD_B = 1024, DUMP_B = 256;
    static_assert(!DIRECT || ((TS + Q) * BPS <= HEAD_B && 3 * TS >= 2 * TS + Q), "the head holds blocks 0..2's new samples");
    static constexpr int RAW_B = DIRECT ? GUARD_B + HEAD_B + DUMP_B : GUARD_B + NDMA16 * 1024 + NDMA4 * 256;
    // FFT exchange area (also the |X|^2 hand-over and the mask estimator's linear spectrum)
    static constexpr int XP_FFT_B = NDFT == 256 ? 4 * PIRIP_XPS256 : NDFT == 512 ? 4480 : 8 * 72 * 8;    // Ndft 128: eight FFTs x (8 groups x 9) complex, two passes
    static constexpr int XP_B = cmax(cmax(XP_FFT_B, NDFT * 4), NDFT == 256 ? 64 * 20 * 4 : NDFT == 128 ? 8 * 136 * 4 : 0);
    // carried between launches in the stream's DemodState::hist block (M * HIST float2 = room to spare): the guard area as it stands
    // (the raw tail, right-aligned), then per tone last frame's phase step and oscillator-table row, then last frame's nin (0: no tail yet)
    static constexpr int TAIL_DW = HIST * BPS / 4;               // dwords of raw tail
    static constexpr int TRAILER_DW = GUARD_B / 4;               // first trailer dword in the state block
    static_assert((HIST * BPS) % 4 == 0 && (GUARD_B - HIST * BPS) % 4 == 0, "tail copies move whole dwords");
    static_assert(GUARD_B + (2 * M + 1) * 4 <= M * HIST * 8, "the state block holds the raw tail and its trailer");
    static constexpr int CHS = (BPS == 2) ? TS : (TS % 8 == 0 ? 8 : TS % 4 == 0 ? 4 : 2);   // samples per correlator chunk (chunk bytes: multiple of 16)
    static constexpr int CH_DW = CHS * BPS / 4;
    static_assert(TS % 2 == 0 && TS % P == 0 && P >= 4, "bad Ts / P");      // (Q = Ts/4 rounds down, as codec2's nin steps do)
    static_assert(NLANES <= kWave, "one lane per symbol block");
    static_assert((TS * BPS) % 16 == 0 && (CHS * BPS) % 16 == 0 && TS % CHS == 0, "block and chunk strides keep 16-byte alignment");
    static_assert((N + Q) / (NDFT / 2) - 1 == NFFT && N / (NDFT / 2) - 1 == NFFT, "numffts must not depend on nin");
    static_assert(NDFT == 256 ? (NFFT >= 4) : NDFT == 512 ? (NFFT % 2 == 0) : (NDFT == 128 && NFFT <= 8),
                  "FFT batches (Ndft = 256: the last batch of 4 may be partial; Ndft = 128: one batch of up to eight)");
    static_assert((NFFT - 1) * (NDFT / 2) + NDFT <= N - Q, "FFT windows stay inside the shortest frame");
    static_assert(M * P <= 48, "window prefix sums are kept in registers");
};

// Per-wave LDS footprint, also used by the launcher to report occupancy
template <class C, int M> constexpr int wave_lds_bytes() { return C::RAW_B + C::XP_B; }

}  // namespace

// FFT_FMA = false: kiss_fft's complex multiply as C computes it without contraction (3 packed ops, every product rounded:
// Sf bit-identical to the oracle). FFT_FMA = true (opt-in, PIRIP_FFT_FMA=1): 2 packed ops with a fused multiply-add, i.e. what
// an aarch64 / -ffp-contract=fast build of codec2 computes; Sf then differs in the last bits (tests report whether f_est / nin /
// bits still match: DESIGN.md 5).
// BAND = 2 or 4 (opt-in, pirip_hip_set_estimator_band_only; Ndft = 256, peak estimator): only the FFT bins the peak search can read --
// bins 0 .. 16 BAND - 1 -- are computed, smoothed and kept. Lane e16 of a 16-lane FFT group ends stage 4 with bins e16 + 16 b' + 64 a' in
// W[4 a' + b']: the band is a' = 0, b' < BAND, so W[0 .. BAND-1] are the only outputs used and everything that feeds only the others
// (BAND = 2: stage 4's butterflies b' = 2, 3 and the outputs 2 and 3 of every stage-3 butterfly; BAND = 4: outputs 1 .. 3 of stage 4's
// butterflies) is dead code the compiler removes. What is left runs the same instructions on the same operands: Sf of the band, f_est
// and every output are bit-identical to the full estimator's.
template <int M, int TS, int P, int NSYM, int NDFT, int FMT, int WPB, int WPS, bool FFT_FMA = false, bool MASK = false, int BAND = 0>
__global__ __launch_bounds__(kWave * WPB, WPS) void fsk_demod_wave_kernel(DemodArgs a_kernarg, int nstreams)
{
    // Where the argument block is read from. By value (the default) every field a frame touches -- eleven table pointers, the output
    // pointers and strides, the 18 stage-2 twiddles, the hand-over block -- is loaded once and held in SGPRs across the frame loop, and 5
    // to 90 of them are spilled into VGPR lanes (v_writelane / v_readlane: VALU issue slots). INPLACE reads the fields through a pointer
    // into the kernarg segment that is made opaque twice per frame: no instance then spills more than 8 SGPRs (most: none) -- and all
    // but three shapes are SLOWER by 0.1 .. 2.7 % (interleaved A/B on one box, profiles/r06_b_ab_kernarg_instances.txt: a scalar load's
    // latency at the point of use costs more than a lane write and a lane read). So: by value, except Ts = 10 (`rtl_fsk -a 100000 -r 10000`,
    // README.md:196: + 1.9 % / + 4.1 %). -DPIRIP_WAVE_KERNARG_INPLACE=0 / 1 force one way for every instance (A/B builds).
#ifdef PIRIP_WAVE_KERNARG_INPLACE
    constexpr bool INPLACE = PIRIP_WAVE_KERNARG_INPLACE != 0;
#else
    constexpr bool INPLACE = TS == 10;
#endif
    static_assert(offsetof(DemodArgs, d) == 0, "the argument block is the first kernel argument");
    auto ap = [&]() {
        if constexpr (INPLACE) return (const __attribute__((address_space(4))) DemodArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        else return (const DemodArgs *)&a_kernarg;
    }();
#define PIRIP_ARGS_REREAD() do { if constexpr (INPLACE) asm volatile("" : "+s"(ap)); } while (0)
#define a (*ap)
#define d (ap->d)
    static_assert(BAND == 0 || ((BAND == 2 || BAND == 4) && NDFT == 256 && !MASK), "band-only estimator: Ndft = 256, peak method, 32 or 64 bins");
    auto cmul = [](v2f x, v2f t) { return FFT_FMA ? rot_step(x, t) : cmul_x(x, t); };
    using C = WaveCfg<M, TS, P, NSYM, NDFT, FMT>;
    constexpr int N = C::N, NMEM = C::NMEM, HIST = C::HIST, STEP = C::STEP, Q = C::Q, BPS = C::BPS;
    constexpr int GUARD_B = C::GUARD_B;

    __shared__ __attribute__((aligned(16))) unsigned char s_raw[WPB][C::RAW_B];
    __shared__ __attribute__((aligned(16))) unsigned char s_xp[WPB][C::XP_B];
    // shared by the block's streams: FFT constants (Ndft = 256: [12 float4 chunks][16 lanes]: Hann samples of the lane's
    // 16 inputs, stage-3/4 twiddles; Ndft = 512: stage-2/3 twiddles [8][16] cf | last-stage twiddles [12][32] cf -- the
    // Hann samples of a lane's 16 inputs are the same for every FFT and live in 16 VGPRs for the whole kernel; the 2 KB
    // they would take here are what lets a third block of the f32-input instance onto a CU)
    constexpr int TAB_F = NDFT == 256 ? 12 * 16 * 4 : NDFT == 512 ? 8 * 16 * 2 + 12 * 32 * 2 : 8 * 16 * 2;
    __shared__ __attribute__((aligned(16))) float s_tab[TAB_F];
    __shared__ __attribute__((aligned(16))) float2 s_tph[P];
    __shared__ __attribute__((aligned(16))) float2 s_tgain[NSYM + 2];    // the upstream fine-timing recursion's gain at each lane's block start
    __shared__ float s_misc[WPB][3];                                     // per stream: snr_est, EbNodB, v_est (observable frames only)
    // SoftOut only (the launcher asks for these 1032 bytes of dynamic LDS just then, so that the other calls keep their occupancy):
    // the ln I0 table, shared by the block's streams -- from global memory each look-up pair cost a cache round trip per frame
    extern __shared__ __attribute__((aligned(16))) float s_lnI0[];

    const int lane0 = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sid = blockIdx.x * WPB + wv;

    if (NDFT == 256) {
        for (int i = threadIdx.x; i < 12 * 16; i += kWave * WPB)
            ((float4 *)s_tab)[i] = ((const float4 *)a.t.fast_tab)[(i & 15) * 12 + (i >> 4)];   // [e16][chunk] -> [chunk][e16]
    } else {
        for (int i = threadIdx.x; i < TAB_F / 4; i += kWave * WPB) ((float4 *)s_tab)[i] = ((const float4 *)(a.t.fast_tab + NDFT))[i];
    }
    if constexpr (P <= 10) if (a.io.soft.llr) {
        if (a.io.soft.llr_map == kLlrRician) { for (int i = threadIdx.x; i < 258; i += kWave * WPB) s_lnI0[i] = a.io.soft.lnI0[i]; }
        else if (threadIdx.x < 20) {
            // codec2's logbesseli0 pieces (fsk_device.hpp: logbesseli0_upstream) as rows (c2, c1, c0, -) of the same LDS area: picked
            // per value by segment index -- as selects the fifteen constants sit in VGPRs for the whole frame loop (+ 12 registers)
            const int sg = threadIdx.x >> 2, cc = threadIdx.x & 3;
            const float c2 = sg == 0 ? 0.226f : sg == 1 ? 0.1245f : sg == 2 ? 0.0288f : sg == 3 ? 0.002f : 0.0f;
            const float c1 = sg == 0 ? 0.0125f : sg == 1 ? 0.2177f : sg == 2 ? 0.6314f : sg == 3 ? 0.9048f : 0.9867f;
            const float c0 = sg == 0 ? -0.0012f : sg == 1 ? -0.108f : sg == 2 ? -0.5645f : sg == 3 ? -1.2997f : -2.2053f;
            s_lnI0[threadIdx.x] = cc == 0 ? c2 : cc == 1 ? c1 : cc == 2 ? c0 : 0.0f;
        }
    }
    if (threadIdx.x < P) s_tph[threadIdx.x] = a.t.tph[threadIdx.x];
    if (threadIdx.x < NSYM + 2) s_tgain[threadIdx.x] = a.t.timing_rec[(threadIdx.x < NSYM + 1 ? threadIdx.x : 0) * P];
    __syncthreads();
    if (sid >= nstreams) return;                         // surplus waves of the last block (no barrier follows)
    if (a.io.seg && a.io.seg[sid].max_frames < 0) return; // capture mode: this slot sits the launch out, its state untouched

    unsigned char *raw = s_raw[wv];
    unsigned char *xpb = s_xp[wv];

    // The guard in front of the staged frame: the raw samples of last frame's tail (this stream's state block), or neutral samples
    // when there is no last frame (state as pirip_hip_reset / fsk_create leave it: integrator memory all zero).
    // Last frame's tone estimates (phase step, oscillator-table row) and length: the old positions are mixed with THOSE.
    uint32_t *st32 = (uint32_t *)(a.s.hist + (size_t)sid * M * HIST);
    int ninp = __builtin_amdgcn_readfirstlane((int)st32[C::TRAILER_DW + 2 * M]);
    uint32_t dthp[M];
    int tixp[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
        dthp[m] = (uint32_t)__builtin_amdgcn_readfirstlane((int)st32[C::TRAILER_DW + m]);
        tixp[m] = __builtin_amdgcn_readfirstlane((int)st32[C::TRAILER_DW + M + m]);
    }
    for (int i = lane0; i < GUARD_B / 4; i += kWave) ((uint32_t *)raw)[i] = ninp ? st32[i] : InFmt<FMT>::NEUTRAL;

    // ---- per-lane estimator state: owned Sf bins -------------------------------------------------------------------
    // Lane-derived indices and addresses are cheap to compute and expensive to keep: hipcc hoists them out of the frame
    // loop and then spills. Every phase therefore starts from its own opaque copy of the lane id.
#define PIRIP_PHASE_LANE(name) int name = lane0; asm volatile("" : "+v"(name))
    constexpr int NOWN = NDFT / kWave;                     // 2 (Ndft 128), 4 (Ndft 256) or 8 (Ndft 512)
    // Sf index (fftshift applied) of owned bin b: Ndft 256: FFT bin e16 + 16 b + 64 grp; Ndft 512: L + 32 (8 hh + b); Ndft 128: lane + 64 b
    // (BAND: every 16-lane group keeps its own copy of bins e16 + 16 b, b < BAND -- the four copies see the same updates; group 0's is
    //  the one the peak search reads and the state keeps)
    constexpr int NB = BAND ? BAND : NOWN;                // Sf registers in use
    auto own_sfi = [](int ln, int b) {
        const int bin = BAND ? ((ln & 15) + 16 * b)
                      : NDFT == 256 ? ((ln & 15) + 16 * b + 64 * (ln >> 4)) : NDFT == 512 ? ((ln & 31) + 32 * (8 * (ln >> 5) + b)) : (ln + 64 * b);
        return (bin + NDFT / 2) & (NDFT - 1);
    };
    float Sf[NOWN];
    // Ndft = 512: Hann samples of this lane's 16 FFT inputs (input L5 + 32 u + 64 t of either half-wave's FFT)
    // Ndft = 128: of inputs l8 + 8 y, y = 0..15, of any of the eight FFTs of a batch
    float hann16[NDFT == 256 ? 1 : 16];
    if constexpr (NDFT == 512) {
#pragma unroll
        for (int i = 0; i < 16; i++) hann16[i] = a.t.fast_tab[(lane0 & 31) + 32 * (i >> 3) + 64 * (i & 7)];
    }
    if constexpr (NDFT == 128) {
#pragma unroll
        for (int i = 0; i < 16; i++) hann16[i] = a.t.fast_tab[(lane0 & 7) + 8 * i];
    }
#pragma unroll
    for (int b = 0; b < NB; b++) Sf[b] = a.s.Sf[(size_t)sid * NDFT + own_sfi(lane0, b)];

    // stream scalars carried frame to frame: only what this kernel updates (the rest of StreamScalars passes through)
    float sc_norm_rx_timing, sc_ppm, sc_SNRest;
    int sc_nin;
    {
        const StreamScalars sc = a.s.scal[sid];
        sc_norm_rx_timing = sc.norm_rx_timing; sc_ppm = sc.ppm; sc_SNRest = sc.SNRest; sc_nin = sc.nin;
        if (lane0 == 0) { s_misc[wv][0] = sc.snr_est; s_misc[wv][1] = sc.EbNodB; s_misc[wv][2] = sc.v_est; }
    }
    // (no oscillator phase is carried from frame to frame: every frame's down-conversion starts at phase 0 before its first new
    //  sample, and the integrator-memory tail handed to the next frame is turned by the phase this frame's oscillators ended at --
    //  the sums see one continuous oscillator per tone exactly as upstream's do, but a stream's state at a frame boundary no longer
    //  depends on every frame before it, which is what lets one long capture be demodulated in independent pieces, capture.hip)

    // a segment of one long capture (capture.hip): own first sample, frame budget and output row; wave-uniform
    int64_t seg_in = 0, out0 = 0, max_frames = a.io.max_frames;
    if (a.io.seg) {
        const SegDesc sd = a.io.seg[sid];
        seg_in = sd.in_off; out0 = sd.out_frame0; max_frames = sd.max_frames;
    }
    // bounds-checked view of this stream's bytes (out-of-range dwords read as 0)
    const uint8_t *in_base = a.io.in + (size_t)sid * a.io.in_stride + (size_t)seg_in * BPS;
    const int64_t nsamp = a.io.nsamp - seg_in;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)in_base, 0, (int)(uint32_t)(BPS * nsamp), 0x00020000);

    int nin = __builtin_amdgcn_readfirstlane(sc_nin);
    int64_t pos = 0, frame = 0;
    // the stream's first frame after fsk_create / reset was demodulated by the exact prologue of this call (fsk_demod_general.hip:
    // fsk_demod_exact0_kernel; P == Ts instances only): start behind it -- its state block, scalars and Sf are what was loaded above
    if (a.io.first) { pos = __builtin_amdgcn_readfirstlane(a.io.first[sid]); frame = pos ? 1 : 0; }
    const int64_t frame_first = frame;
    int last_freqi0 = 0, last_freqi1 = 0, last_freqi2 = 0, last_freqi3 = 0;   // tone bins of the last frame (uniform)

    // LDS-DMA of the frame superset [p0, p0 + N + Q) to raw + GUARD_B (lane-linear: 16 or 4 bytes per lane per instruction)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto dma_frame = [&](int64_t p0) {
        const uint32_t goff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p0 * BPS));
        if constexpr (C::DIRECT) {
            // the head behind the guard (one 16-byte-per-lane copy covers it), then one dword of every 128-byte line of the superset
            // into the dump row: the lines are on their way to L2 while this frame's window sums, timing and decisions run
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(raw + GUARD_B), 16, lane0 * 16, goff, 0, 0);
#if PIRIP_F32_PREFETCH
#pragma unroll
            for (int i = 0; i < (C::SUP_B + 8191) / 8192; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(raw + GUARD_B + C::HEAD_B), 4, lane0 * 128, goff + i * 8192, 0, 0);
#endif
            return;
        }
#pragma unroll
        for (int i = 0; i < C::NDMA16; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(raw + GUARD_B + i * 1024), 16, lane0 * 16, goff + i * 1024, 0, 0);
#pragma unroll
        for (int i = 0; i < C::NDMA4; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(raw + GUARD_B + C::NDMA16 * 1024 + i * 256), 4, lane0 * 4,
                                                     goff + C::NDMA16 * 1024 + i * 256, 0, 0);
    };
    // SoftOut: hard-decision words are assembled across frames (Nbits is not a multiple of 32): bits waiting for their word
    constexpr bool SOFT_OK = P <= 10;                      // every shape rtl_fsk --code and BASELINE config 4 use; not fsk_demod -p 24
    uint32_t soft_carry = 0;
    int soft_cb = 0, soft_w = 0;
    uint32_t *words_o = nullptr;
    if constexpr (SOFT_OK) if (a.io.soft.llr) { words_o = a.io.soft.words + (size_t)sid * a.io.soft.words_stride + (a.io.soft.bit0 >> 5); }
    // top nb bits of W (first bit in the MSB, the rest zero) behind the bits already waiting; wave-uniform integer code
    auto soft_append = [&](uint32_t W, int nb) {
        const uint32_t merged = soft_carry | (soft_cb ? (W >> soft_cb) : W);
        if (soft_cb + nb >= 32) {
            if (lane0 == 0) words_o[soft_w] = merged;
            soft_w++;
            soft_carry = soft_cb ? (W << (32 - soft_cb)) : 0u;
            soft_cb = soft_cb + nb - 32;
        } else { soft_carry = merged; soft_cb += nb; }
    };
    constexpr int NBITS = NSYM * (M == 2 ? 1 : 2);
    wave_lds_sync();
    dma_frame(pos);
    PIRIP_T_DECL;

    while (frame < max_frames && pos + nin <= nsamp) {
        PIRIP_ARGS_REREAD();                              // (argument fields are re-read from the kernarg segment from here on: see the top)
        const int nold = NMEM - nin;                       // 2 Ts -/0/+ Ts/4 (uniform)
        // the staged frame has landed (LDS-DMA is ordered only by this wave's vmcnt; the wave is its only reader)
        PIRIP_T_MARK(7);                                   // loop overhead / previous frame's tail
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
        PIRIP_T_MARK(0);                                   // waiting for the staged frame
        const unsigned char *smp = raw + GUARD_B;          // new sample i of the frame at smp + i * BPS
        const uint32_t goff_frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pos * BPS));
        // one FFT input: staged formats read LDS at the address, the unstaged one the same offset of the frame in global memory
        auto fft_in = [&](const unsigned char *p) -> v2f {
            if constexpr (C::DIRECT && BPS == 8) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(p - smp), (int)goff_frame, 0);
                return __builtin_bit_cast(v2f, v);
            } else if constexpr (C::DIRECT) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(p - smp), (int)goff_frame, 0);
                return v2f{cvt_s16((float)(short)(v & 0xffffu)), cvt_s16((float)((int)v >> 16))};
            } else return lds_sample<FMT>(p);
        };
        // ================= a-5: frequency estimator =================================================================
        const v2f ktc{d.one_minus_tc, d.tc};
        if constexpr (NDFT == 256) {
            PIRIP_PHASE_LANE(lane);
            const int grp = lane >> 4, e16 = lane & 15;    // 4 FFTs x 16 lanes
            const float4 *ftab = (const float4 *)s_tab + e16;
            constexpr int XPS = PIRIP_XPS256;              // bytes per FFT group: 16 rows x 17 cf
            // this lane's FFT constants, fetched once per frame (48 VGPRs that are free until the correlator starts):
            // chunks 0..3 Hann samples of its 16 inputs, 4..5 stage-3 twiddles, 5..11 stage-4 twiddles
            float4 tabv[12];
#pragma unroll
            for (int i = 0; i < 12; i++) tabv[i] = ftab[16 * i];
#pragma unroll 1
            for (int bt = 0; bt < (C::NFFT + 3) / 4; bt++) {
                int jj = 4 * bt + grp;                     // this 16-lane group's FFT
                if (C::NFFT % 4) jj = jj < C::NFFT ? jj : C::NFFT - 1;   // partial last batch: idle groups repeat the last FFT, unused
                const int ga = e16 >> 2, gb = e16 & 3;
                const int base = ga + 4 * gb;
                const unsigned char *src = smp + BPS * ((NDFT / 2) * jj + base);
                v2f W[16];
                v2f hann2[8];                                  // this lane's 16 window samples as register pairs
                {
                    const float4 h0 = tabv[0], h1 = tabv[1], h2 = tabv[2], h3 = tabv[3];
                    hann2[0] = v2f{h0.x, h0.y}; hann2[1] = v2f{h0.z, h0.w}; hann2[2] = v2f{h1.x, h1.y}; hann2[3] = v2f{h1.z, h1.w};
                    hann2[4] = v2f{h2.x, h2.y}; hann2[5] = v2f{h2.z, h2.w}; hann2[6] = v2f{h3.x, h3.y}; hann2[7] = v2f{h3.z, h3.w};
                }
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const v2f x = fft_in(src + 16 * BPS * t);
                    const int c = t & 3, dd = t >> 2;
                    W[4 * c + dd] = (t & 1) ? scale_hi(hann2[t >> 1], x) : scale_lo(hann2[t >> 1], x);
                }
                // stage 1 (m=1): trivial twiddles (x (1,-0): identical up to the sign of zero)
#pragma unroll
                for (int c = 0; c < 4; c++) bfly4(W[4 * c], W[4 * c + 1], W[4 * c + 2], W[4 * c + 3]);
                // stage 2 (m=4, fstride 16)
                bfly4(W[0], W[4], W[8], W[12]);
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    v2f f1 = cmul(W[4 + k], v2f{a.tw_s2[6 * (k - 1) + 0], a.tw_s2[6 * (k - 1) + 1]});
                    v2f f2 = cmul(W[8 + k], v2f{a.tw_s2[6 * (k - 1) + 2], a.tw_s2[6 * (k - 1) + 3]});
                    v2f f3 = cmul(W[12 + k], v2f{a.tw_s2[6 * (k - 1) + 4], a.tw_s2[6 * (k - 1) + 5]});
                    bfly4(W[k], f1, f2, f3);
                    W[4 + k] = f1; W[8 + k] = f2; W[12 + k] = f3;
                }
                // 16x16 transpose inside the 16-lane group: rows of 17 cf (conflict-free both ways, and every access is
                // one base register + an immediate offset; an XOR swizzle would save 512 B per wave but costs 32 address
                // registers that the compiler hoists out of the frame loop); 2176 B per group puts neighbouring groups in
                // opposite bank halves
                {
                    float2 *xp = (float2 *)(xpb + grp * XPS);
#pragma unroll
                    for (int e = 0; e < 16; e++) xp[e16 * 17 + e] = make_float2(W[e].x, W[e].y);
                    wave_lds_sync();
#pragma unroll
                    for (int g = 0; g < 16; g++) { const float2 v = xp[g * 17 + e16]; W[g] = v2f{v.x, v.y}; }
                }
                // stage 3 (m=16, fstride 4): twiddles of this lane (chunks 4..5 of its table row: 3 cf)
                {
                    const float4 c4 = tabv[4], c5 = tabv[5];
                    const v2f tw3[3] = {v2f{c4.x, c4.y}, v2f{c4.z, c4.w}, v2f{c5.x, c5.y}};
#pragma unroll
                    for (int aa = 0; aa < 4; aa++) {
                        v2f f1 = cmul(W[4 * aa + 1], tw3[0]);
                        v2f f2 = cmul(W[4 * aa + 2], tw3[1]);
                        v2f f3 = cmul(W[4 * aa + 3], tw3[2]);
                        bfly4(W[4 * aa], f1, f2, f3);
                        W[4 * aa + 1] = f1; W[4 * aa + 2] = f2; W[4 * aa + 3] = f3;
                    }
                }
                // stage 4 (m=64, fstride 1): 3 cf per b' from floats 22.. of the table row, fetched as they are used
#pragma unroll
                for (int b = 0; b < (BAND ? BAND : 4); b++) {
                    const float *trow = (const float *)(ftab);            // this lane's row, float index f at chunk f/4, component f%4
                    auto tf = [&](int f) { const float4 c = tabv[f >> 2]; return (f & 3) == 0 ? c.x : (f & 3) == 1 ? c.y : (f & 3) == 2 ? c.z : c.w; };
                    (void)trow;
                    const v2f t1{tf(22 + 6 * b), tf(23 + 6 * b)}, t2{tf(24 + 6 * b), tf(25 + 6 * b)}, t3{tf(26 + 6 * b), tf(27 + 6 * b)};
                    v2f f1 = cmul(W[4 + b], t1);
                    v2f f2 = cmul(W[8 + b], t2);
                    v2f f3 = cmul(W[12 + b], t3);
                    bfly4(W[b], f1, f2, f3);
                    W[4 + b] = f1; W[8 + b] = f2; W[12 + b] = f3;
                }
                // |X|^2 of bin e16 + 16 b' + 64 a' sits in W[4a'+b']; hand each to the lane owning the bin
                wave_lds_sync();
                if constexpr (BAND != 0) {
                    // this lane's BAND band bins of ITS FFT: square roots where they are (64 lanes x BAND = the batch's values), then the four
                    // FFTs' magnitudes of a bin meet in time order on every lane with that e16
                    float m2b[BAND], rb[BAND];
                    unsigned kmin = 0xffffffffu;
#pragma unroll
                    for (int b = 0; b < BAND; b++) { m2b[b] = mag2(W[b]); kmin = umin2(kmin, fbits(m2b[b])); }
                    if (__all(kmin >= kSqrtLo)) {
#pragma unroll
                        for (int b = 0; b < BAND; b++) rb[b] = sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(m2b[b]);
                    } else {
                        kmin = 0xffffffffu;
#pragma unroll
                        for (int b = 0; b < BAND; b++) kmin = umin2(kmin, sqrt_key(m2b[b]));
                        if (__all(kmin >= kSqrtLo - 1u)) {
#pragma unroll
                            for (int b = 0; b < BAND; b++) rb[b] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(m2b[b]);
                        } else {
#pragma unroll
                            for (int b = 0; b < BAND; b++) rb[b] = sqrtf(m2b[b]);
                        }
                    }
                    float2 *mx = (float2 *)xpb;                              // [pair][FFT group][e16]
#pragma unroll
                    for (int b = 0; b < BAND; b += 2) mx[(b / 2) * 64 + lane] = make_float2(rb[b], rb[b + 1]);
                    wave_lds_sync();
                    float2 rt[BAND / 2][4];
#pragma unroll
                    for (int b = 0; b < BAND; b += 2)
#pragma unroll
                        for (int g2 = 0; g2 < 4; g2++) rt[b / 2][g2] = mx[(b / 2) * 64 + g2 * 16 + e16];
#pragma unroll
                    for (int b = 0; b < BAND; b += 2) {
                        v2f sp{Sf[b], Sf[b + 1]};
#pragma unroll
                        for (int g2 = 0; g2 < 4; g2++)
                            if (C::NFFT % 4 == 0 || 4 * bt + g2 < C::NFFT) sp = smooth2(sp, v2f{rt[b / 2][g2].x, rt[b / 2][g2].y}, ktc);
                        Sf[b] = sp.x; Sf[b + 1] = sp.y;
                    }
                    wave_lds_sync();
                } else {
                    float *mx = (float *)xpb;
                    float4 *row = (float4 *)(mx + lane * 20);
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++)
                        row[q4] = make_float4(mag2(W[4 * q4 + 0]), mag2(W[4 * q4 + 1]), mag2(W[4 * q4 + 2]), mag2(W[4 * q4 + 3]));
                    wave_lds_sync();
                    float4 m2[4];
                    unsigned kmin = 0xffffffffu;
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2++) {
                        m2[g2] = *(const float4 *)(mx + (g2 * 16 + e16) * 20 + 4 * grp);
                        kmin = umin3(umin3(kmin, fbits(m2[g2].x), fbits(m2[g2].y)), fbits(m2[g2].z), fbits(m2[g2].w));
                    }
                    // square roots first (branch on the wave-uniform range test), then the smoothing in time order
                    // (keep this shape: with the Sf updates written inside both branches hipcc 7.2 hoisted Sf[0]*(1-tc) above
                    //  the branch onto a register it had just reused for kmin -- wrong Sf[0] in every batch; the bit-exact Sf
                    //  parity test catches it)
                    float4 rt[4];
                    if (__all(kmin >= kSqrtLo)) {                          // every |X|^2 >= 2^-96: no zero guard in the roots
#pragma unroll
                        for (int g2 = 0; g2 < 4; g2++)
                            rt[g2] = make_float4(sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(m2[g2].x), sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(m2[g2].y),
                                                 sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(m2[g2].z), sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(m2[g2].w));
                    } else {
                        kmin = 0xffffffffu;
#pragma unroll
                        for (int g2 = 0; g2 < 4; g2++)
                            kmin = umin3(umin3(kmin, sqrt_key(m2[g2].x), sqrt_key(m2[g2].y)), sqrt_key(m2[g2].z), sqrt_key(m2[g2].w));
                        if (__all(kmin >= kSqrtLo - 1u)) {                  // zeros among them (a silent input): the guarded form
#pragma unroll
                            for (int g2 = 0; g2 < 4; g2++)
                                rt[g2] = make_float4(sqrt_rn_normal<FMT != PIRIP_IN_CF32>(m2[g2].x), sqrt_rn_normal<FMT != PIRIP_IN_CF32>(m2[g2].y),
                                                     sqrt_rn_normal<FMT != PIRIP_IN_CF32>(m2[g2].z), sqrt_rn_normal<FMT != PIRIP_IN_CF32>(m2[g2].w));
                        } else {
#pragma unroll
                            for (int g2 = 0; g2 < 4; g2++) {
                                rt[g2] = make_float4(sqrtf(m2[g2].x), sqrtf(m2[g2].y), sqrtf(m2[g2].z), sqrtf(m2[g2].w));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    {
                        v2f s01{Sf[0], Sf[1]}, s23{Sf[2], Sf[3]};
#pragma unroll
                        for (int g2 = 0; g2 < 4; g2++) {
                            if (C::NFFT % 4 == 0 || 4 * bt + g2 < C::NFFT) {
                                s01 = smooth2(s01, v2f{rt[g2].x, rt[g2].y}, ktc);
                                s23 = smooth2(s23, v2f{rt[g2].z, rt[g2].w}, ktc);
                            }
                        }
                        Sf[0] = s01.x; Sf[1] = s01.y; Sf[2] = s23.x; Sf[3] = s23.y;
                    }
                    wave_lds_sync();
                }
            }
        } else if constexpr (NDFT == 128) {
            // ---- Ndft = 128: kiss_fft factors 4,4,4,2 (executed leaf first: radix-2 m=1, radix-4 m=2, 8, 32). All of the frame's
            // FFTs (5 at Ts = 8, 6 at Ts = 10) in ONE batch: eight lanes per FFT, 16 points per lane -- the inner part of the
            // Ndft = 512 dataflow (one exchange, no m = 128 level):
            //   phase 1  lane l8: the two 8-point leaf groups g = l8 + 8 u (g = q0 + 4 q1) fed by inputs g + 16 q2 + 64 q3 (levels m=1, m=2)
            //   phase 2  lane r = l8: slots q0*32 + q1*8 + r for all (q0, q1) (levels m=8, m=32); V[i] ends up as bin r + 8 i
            // Exchange in two passes (the leaf groups with u = 0, then u = 1: half the LDS, which is what decides how many streams
            // fit a CU at these frame sizes): slot (f*72 + (g - 8u)*9 + r) complex -- groups 9 apart and FFTs 72 apart put a write
            // instruction's 64 lanes on all banks twice and a read instruction's likewise (the minimum for 64 x 8 bytes).
            PIRIP_PHASE_LANE(lane);
            const int f8 = lane >> 3, l8 = lane & 7;
            const int jj = f8 < C::NFFT ? f8 : C::NFFT - 1;       // idle groups repeat the last FFT; their rows are never read
            const float2 *s_p2 = (const float2 *)s_tab;
            float2 *xf = (float2 *)xpb + f8 * 72;
            v2f V[16];                                          // V[4 q0 + q1] = slot q0*32 + q1*8 + r, group g = q0 + 4 q1 (pass u brings q1 = 2u, 2u + 1)
            {
                const unsigned char *src = smp + BPS * ((NDFT / 2) * jj + l8);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    v2f S[8];
                    {
                        v2f Wt[8];
#pragma unroll
                        for (int t = 0; t < 8; t++) {          // t = q2 + 4 q3: input l8 + 8 (u + 2 q2 + 8 q3)
                            const int y = u + 2 * (t & 3) + 8 * (t >> 2);
                            const v2f x = fft_in(src + BPS * 8 * y);
                            const float hn = hann16[y];
                            Wt[t] = v2f{hn * x.x, hn * x.y};
                        }
#pragma unroll
                        for (int q2 = 0; q2 < 4; q2++) { S[2 * q2] = Wt[q2] + Wt[q2 + 4]; S[2 * q2 + 1] = Wt[q2] - Wt[q2 + 4]; }
                    }
                    // radix-4, m = 2, fstride 16: k = 0 trivial; k = 1 with tw[16], tw[32], tw[48] (uniform constants)
                    bfly4(S[0], S[2], S[4], S[6]);
                    {
                        v2f f1 = cmul(S[3], v2f{a.tw_s2[0], a.tw_s2[1]});
                        v2f f2 = cmul(S[5], v2f{a.tw_s2[2], a.tw_s2[3]});
                        v2f f3 = cmul(S[7], v2f{a.tw_s2[4], a.tw_s2[5]});
                        bfly4(S[1], f1, f2, f3);
                        S[3] = f1; S[5] = f2; S[7] = f3;
                    }
                    if (u == 1) wave_lds_sync();            // pass 0's reads are done
                    float2 *wr = xf + l8 * 9;
#pragma unroll
                    for (int r = 0; r < 8; r++) wr[r] = make_float2(S[r].x, S[r].y);
                    wave_lds_sync();
#pragma unroll
                    for (int q0 = 0; q0 < 4; q0++)
#pragma unroll
                        for (int ql = 0; ql < 2; ql++) { const float2 v = xf[(q0 + 4 * ql) * 9 + l8]; V[4 * q0 + 2 * u + ql] = v2f{v.x, v.y}; }
                }
            }
            // radix-4, m = 8, fstride 4: over q1 for each q0, k = r
            {
                const float2 t1 = s_p2[l8 * 16 + 0], t2 = s_p2[l8 * 16 + 1], t3 = s_p2[l8 * 16 + 2];
#pragma unroll
                for (int q0 = 0; q0 < 4; q0++) {
                    v2f f1 = cmul(V[4 * q0 + 1], v2f{t1.x, t1.y});
                    v2f f2 = cmul(V[4 * q0 + 2], v2f{t2.x, t2.y});
                    v2f f3 = cmul(V[4 * q0 + 3], v2f{t3.x, t3.y});
                    bfly4(V[4 * q0], f1, f2, f3);
                    V[4 * q0 + 1] = f1; V[4 * q0 + 2] = f2; V[4 * q0 + 3] = f3;
                }
            }
            // radix-4, m = 32, fstride 1: over q0 for each j1, k = r + 8 j1; output j0 is bin k + 32 j0 = r + 8 (j1 + 4 j0)
#pragma unroll
            for (int j1 = 0; j1 < 4; j1++) {
                const float2 t1 = s_p2[l8 * 16 + 3 + 3 * j1], t2 = s_p2[l8 * 16 + 4 + 3 * j1], t3 = s_p2[l8 * 16 + 5 + 3 * j1];
                v2f f1 = cmul(V[4 + j1], v2f{t1.x, t1.y});
                v2f f2 = cmul(V[8 + j1], v2f{t2.x, t2.y});
                v2f f3 = cmul(V[12 + j1], v2f{t3.x, t3.y});
                bfly4(V[j1], f1, f2, f3);
                V[4 + j1] = f1; V[8 + j1] = f2; V[12 + j1] = f3;
            }
            // |X|^2 of bin r + 8 i to row f8 of a [8][136] float array (rows 8 banks apart: conflict-free), then every lane
            // collects its two bins (lane, lane + 64) from the frame's FFTs in time order
            wave_lds_sync();
            {
                float *mx = (float *)xpb;
#pragma unroll
                for (int i = 0; i < 16; i++) mx[f8 * 136 + l8 + 8 * i] = mag2(V[i]);
                wave_lds_sync();
                float A[C::NFFT], B[C::NFFT];
                unsigned kmin = 0xffffffffu;
#pragma unroll
                for (int j = 0; j < C::NFFT; j++) {
                    A[j] = mx[j * 136 + lane]; B[j] = mx[j * 136 + lane + 64];
                    kmin = umin3(kmin, fbits(A[j]), fbits(B[j]));
                }
                if (__all(kmin >= kSqrtLo)) {
#pragma unroll
                    for (int j = 0; j < C::NFFT; j++) { A[j] = sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(A[j]); B[j] = sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(B[j]); }
                } else {
                    kmin = 0xffffffffu;
#pragma unroll
                    for (int j = 0; j < C::NFFT; j++) kmin = umin3(kmin, sqrt_key(A[j]), sqrt_key(B[j]));
                    if (__all(kmin >= kSqrtLo - 1u)) {
#pragma unroll
                        for (int j = 0; j < C::NFFT; j++) { A[j] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(A[j]); B[j] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(B[j]); }
                    } else {
#pragma unroll
                        for (int j = 0; j < C::NFFT; j++) { A[j] = sqrtf(A[j]); B[j] = sqrtf(B[j]); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
                v2f sp{Sf[0], Sf[1]};
#pragma unroll
                for (int j = 0; j < C::NFFT; j++) sp = smooth2(sp, v2f{A[j], B[j]}, ktc);
                Sf[0] = sp.x; Sf[1] = sp.y;
                wave_lds_sync();
            }
        } else {
            // ---- Ndft = 512: kiss_fft factors 4,4,4,4,2 (executed leaf first: radix-2 m=1, radix-4 m=2, 8, 32, 128).
            // Two FFTs per batch, one per half wave; 16 points per lane:
            //   phase 1  lane L: the two 8-point leaf groups fed by inputs L + 64 t and L + 32 + 64 t (levels m=1, m=2)
            //   phase 2  lane (q0 = L>>3, r2 = L&7): slots q0*128 + 8x + r2, x = 0..15 (levels m=8, m=32)
            //   phase 3  lane L: bins L + 32 jj (+128 j0)  (level m=128)
            PIRIP_PHASE_LANE(lane);
            const int hh = lane >> 5, L5 = lane & 31;      // 2 FFTs x 32 lanes
            const float2 *s_p2 = (const float2 *)s_tab;
            const float2 *s_p3 = (const float2 *)(s_tab + 8 * 16 * 2);
            const int q0w = L5 & 3, q1w = (L5 >> 2) & 3, q2w = L5 >> 4;          // phase-1 lane as group digits
            const int q0r = L5 >> 3, r2 = L5 & 7;                                // phase-2 lane
            float2 *x1 = (float2 *)xpb + hh * 280;                               // exchange 1: slot (q0*9 + q1*2 + q2l)*8 + (r ^ q1)
            float2 *x2 = (float2 *)xpb + hh * 160;                               // exchange 2: slot q0*40 + k
#pragma unroll 1
            for (int bt = 0; bt < C::NFFT / 2; bt++) {
                const int jj = 2 * bt + hh;
                const unsigned char *src = smp + BPS * ((NDFT / 2) * jj + L5);
                v2f V[16];                                                      // phase 2/3 working set
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    v2f S[8];
                    {
                        v2f Wt[8];
#pragma unroll
                        for (int t = 0; t < 8; t++) {
                            const v2f x = fft_in(src + BPS * (32 * u + 64 * t));
                            const float hn = hann16[8 * u + t];
                            Wt[t] = v2f{hn * x.x, hn * x.y};
                        }
                        // radix-2 leaves (m = 1, twiddle (1,-0): identical up to the sign of zero): inputs t = q3, q3 + 4
#pragma unroll
                        for (int q3 = 0; q3 < 4; q3++) { S[2 * q3] = Wt[q3] + Wt[q3 + 4]; S[2 * q3 + 1] = Wt[q3] - Wt[q3 + 4]; }
                    }
                    // radix-4, m = 2, fstride 64: k = 0 trivial; k = 1 with tw[64], tw[128], tw[192] (uniform constants)
                    bfly4(S[0], S[2], S[4], S[6]);
                    {
                        v2f f1 = cmul(S[3], v2f{a.tw_s2[0], a.tw_s2[1]});
                        v2f f2 = cmul(S[5], v2f{a.tw_s2[2], a.tw_s2[3]});
                        v2f f3 = cmul(S[7], v2f{a.tw_s2[4], a.tw_s2[5]});
                        bfly4(S[1], f1, f2, f3);
                        S[3] = f1; S[5] = f2; S[7] = f3;
                    }
                    // exchange 1, pass u: groups with q2 in {2u, 2u+1}
                    if (u == 1) wave_lds_sync();
                    {
                        float2 *wr = x1 + (q0w * 9 + q1w * 2 + q2w) * 8;
#pragma unroll
                        for (int r = 0; r < 8; r++) wr[r ^ q1w] = make_float2(S[r].x, S[r].y);
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++)
#pragma unroll
                        for (int q2l = 0; q2l < 2; q2l++) {
                            const float2 v = x1[(q0r * 9 + q1 * 2 + q2l) * 8 + (r2 ^ q1)];
                            V[4 * q1 + 2 * u + q2l] = v2f{v.x, v.y};
                        }
                }
                // radix-4, m = 8, fstride 16: over q2 for each q1, k = r2
                {
                    const float2 t1 = s_p2[r2 * 16 + 0], t2 = s_p2[r2 * 16 + 1], t3 = s_p2[r2 * 16 + 2];
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++) {
                        v2f f1 = cmul(V[4 * q1 + 1], v2f{t1.x, t1.y});
                        v2f f2 = cmul(V[4 * q1 + 2], v2f{t2.x, t2.y});
                        v2f f3 = cmul(V[4 * q1 + 3], v2f{t3.x, t3.y});
                        bfly4(V[4 * q1], f1, f2, f3);
                        V[4 * q1 + 1] = f1; V[4 * q1 + 2] = f2; V[4 * q1 + 3] = f3;
                    }
                }
                // radix-4, m = 32, fstride 4: over q1 for each j2, k = r2 + 8 j2
#pragma unroll
                for (int j2 = 0; j2 < 4; j2++) {
                    const float2 t1 = s_p2[r2 * 16 + 3 + 3 * j2], t2 = s_p2[r2 * 16 + 4 + 3 * j2], t3 = s_p2[r2 * 16 + 5 + 3 * j2];
                    v2f f1 = cmul(V[4 + j2], v2f{t1.x, t1.y});
                    v2f f2 = cmul(V[8 + j2], v2f{t2.x, t2.y});
                    v2f f3 = cmul(V[12 + j2], v2f{t3.x, t3.y});
                    bfly4(V[j2], f1, f2, f3);
                    V[4 + j2] = f1; V[8 + j2] = f2; V[12 + j2] = f3;
                }
                // exchange 2 in four passes (one per j1): V[4 j1 + j2] = slot q0*128 + r2 + 8 j2 + 32 j1  ->  Y[4 q0' + j1] at lane k
                v2f Y[16];
                wave_lds_sync();
#pragma unroll
                for (int j1 = 0; j1 < 4; j1++) {
                    if (j1) wave_lds_sync();
#pragma unroll
                    for (int j2 = 0; j2 < 4; j2++) x2[q0r * 40 + r2 + 8 * j2] = make_float2(V[4 * j1 + j2].x, V[4 * j1 + j2].y);
                    wave_lds_sync();
#pragma unroll
                    for (int q0 = 0; q0 < 4; q0++) { const float2 v = x2[q0 * 40 + L5]; Y[4 * q0 + j1] = v2f{v.x, v.y}; }
                }
                // radix-4, m = 128, fstride 1: over q0 for each jj, k0 = L + 32 jj; output j0 is bin k0 + 128 j0
                float mag[16];                                                  // mag[jj + 4 j0] = |X[L + 32 (jj + 4 j0)]|^2
#pragma unroll
                for (int j1 = 0; j1 < 4; j1++) {
                    const float2 t1 = s_p3[(3 * j1 + 0) * 32 + L5], t2 = s_p3[(3 * j1 + 1) * 32 + L5], t3 = s_p3[(3 * j1 + 2) * 32 + L5];
                    v2f f0 = Y[j1];
                    v2f f1 = cmul(Y[4 + j1], v2f{t1.x, t1.y});
                    v2f f2 = cmul(Y[8 + j1], v2f{t2.x, t2.y});
                    v2f f3 = cmul(Y[12 + j1], v2f{t3.x, t3.y});
                    bfly4(f0, f1, f2, f3);
                    mag[j1] = mag2(f0); mag[j1 + 4] = mag2(f1); mag[j1 + 8] = mag2(f2); mag[j1 + 12] = mag2(f3);
                }
                // Lane (hh, L) owns bins L + 32 u for u in [8 hh, 8 hh + 8). v_permlane32_swap exchanges the halves'
                // foreign eight: afterwards A = this bin's |X|^2 in the batch's first FFT (in time), B = in its second.
                float A[8], B[8];
                unsigned kmin = 0xffffffffu;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    // (inline asm: with hipcc 7.2 both elements of __builtin_amdgcn_permlane32_swap's result read back as the
                    //  first one -- tools/compiler_checks.hip; s_nop 1 covers the VALU-write -> permlane-read hazard)
                    A[u] = mag[u]; B[u] = mag[u + 8];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(A[u]), "+v"(B[u]));
                    kmin = umin3(kmin, fbits(A[u]), fbits(B[u]));
                }
                if (__all(kmin >= kSqrtLo)) {
#pragma unroll
                    for (int u = 0; u < 8; u++) { A[u] = sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(A[u]); B[u] = sqrt_rn_normal<FMT != PIRIP_IN_CF32, true>(B[u]); }
                } else {
                    kmin = 0xffffffffu;
#pragma unroll
                    for (int u = 0; u < 8; u++) kmin = umin3(kmin, sqrt_key(A[u]), sqrt_key(B[u]));
                    if (__all(kmin >= kSqrtLo - 1u)) {
#pragma unroll
                        for (int u = 0; u < 8; u++) { A[u] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(A[u]); B[u] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(B[u]); }
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; u++) { A[u] = sqrtf(A[u]); B[u] = sqrtf(B[u]); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    v2f sp{Sf[u], Sf[u + 1]};
                    sp = smooth2(sp, v2f{A[u], A[u + 1]}, ktc);
                    sp = smooth2(sp, v2f{B[u], B[u + 1]}, ktc);
                    Sf[u] = sp.x; Sf[u + 1] = sp.y;
                }
                wave_lds_sync();
            }
        }

        __builtin_amdgcn_sched_barrier(0);
        PIRIP_T_MARK(1);                                   // estimator FFTs
        // ---- tone estimate: M spectral peaks (blank +-f_zero bins around each, ascending order), or -- MASK, codec2's
        //      second estimator, `--mask` -- the position of a comb of 3-bin teeth at multiples of the tone spacing
        int freqi[M];              // peak method: tone bins
#pragma unroll
        for (int m = 0; m < M; m++) freqi[m] = 0;
        int bb = 0;                // mask method: comb position (index into the shifted spectrum)
        if constexpr (MASK) {
            PIRIP_PHASE_LANE(lane);
            // Sf lives in registers, bins interleaved across lanes: lay it out linearly (the FFT exchange area is free now)
            float *sfl = (float *)xpb;
#pragma unroll
            for (int b = 0; b < NOWN; b++) sfl[own_sfi(lane, b)] = Sf[b];
            wave_lds_sync();
            float best = 0.0f; int ib = d.est_st;          // lanes that find nothing keep (0, est_st): smallest index wins ties
            for (int b = d.est_st + lane; b < d.est_en - d.mask_len; b += kWave) {
                float corr = 0.0f;
                for (int k = 0; k < d.n_teeth; k++) corr += sfl[b + a.t.teeth[k]];    // tooth sums in ascending order
                if (corr > best) { best = corr; ib = b; }
            }
            wargmax(best, ib);
            bb = ib;
            wave_lds_sync();
        } else {
            PIRIP_PHASE_LANE(lane);
            float w[NB];
            int sfi[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) {
                sfi[b] = (BAND && lane >= 16) ? -1 : own_sfi(lane, b);                                 // (BAND: group 0's copy competes)
                // the search range is tested ONCE per frame, not once per tone: a bin outside it competes with 0, which the strict
                // comparison against best >= 0 never selects -- the same winners as testing the range in every tone's loop
                w[b] = (sfi[b] >= d.est_st && sfi[b] < d.est_en) ? Sf[b] : 0.0f;
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                float best = 0.0f; int ib = 0;
#pragma unroll
                for (int b = 0; b < NB; b++)
                    if (w[b] > best) { best = w[b]; ib = sfi[b]; }
                wargmax(best, ib);
                int f_min = ib - d.f_zero; f_min = f_min < 0 ? 0 : f_min;
                int f_max = ib + d.f_zero; f_max = f_max > NDFT ? NDFT : f_max;
#pragma unroll
                for (int b = 0; b < NB; b++) if (sfi[b] >= f_min && sfi[b] < f_max) w[b] = 0.0f;
                freqi[m] = ib - NDFT / 2;
            }
#pragma unroll
            for (int x = 1; x < M; x++)
#pragma unroll
                for (int y = x; y > 0; y--)
                    if (freqi[y] < freqi[y - 1]) { const int t = freqi[y]; freqi[y] = freqi[y - 1]; freqi[y - 1] = t; }
        }
        // per tone: phase step per sample (2^32 = one turn) and the row of the oscillator-model tables
        constexpr int LOG2N = NDFT == 128 ? 7 : NDFT == 256 ? 8 : 9;
        uint32_t dthv[M];
        int tix[M];
#pragma unroll
        for (int m = 0; m < M; m++) {
            if constexpr (MASK) { dthv[m] = a.t.mask_dtheta[bb * M + m]; tix[m] = bb * M + m; }
            else { dthv[m] = (uint32_t)freqi[m] << (32 - LOG2N); tix[m] = freqi[m] + NDFT / 2; }
        }

        // exp(+j th), th in 2^-32 turns: the twiddle table, and for the comb's tones (not on FFT bins) a turn by the phase bits below the table index
        auto phasor = [&](uint32_t th) {
            const float2 w = a.t.tw[th >> (32 - LOG2N)];   // exp(-j theta)
            float pc = w.x, ps = -w.y;
            if constexpr (MASK) {
                const float bl = (float)(th & ((1u << (32 - LOG2N)) - 1u)) * 1.4629180792671596e-9f;   // 2 pi / 2^32
                const float b2 = bl * bl;
                const float cb = 1.0f - b2 * (0.5f - b2 * (1.0f / 24.0f));
                const float sb = bl * (1.0f - b2 * ((1.0f / 6.0f) - b2 * (1.0f / 120.0f)));
                const float c2 = pc * cb - ps * sb, s2 = ps * cb + pc * sb;
                pc = c2; ps = s2;
            }
            return make_float2(pc, ps);
        };
        __builtin_amdgcn_sched_barrier(0);
        PIRIP_T_MARK(2);                                   // peak pick
        // ================= a-6: down-convert this lane's Ts samples with every tone, prefix sums ======================
        v2f fi[M][P];              // prefix sums at the window starts, then f_int of this lane's P window starts (VGPR pairs)
        v2f tot[M];
        {
            PIRIP_PHASE_LANE(lane);
            const int lb = lane < C::NLANES ? lane : C::NLANES - 1;          // idle lanes shadow the last block (results unused)
            // this block's raw bytes: integrator position j = Ts*lb + k is new sample j - nold (negative: neutral guard)
            const int boff = GUARD_B + (TS * lb - nold) * BPS;
            const int al = (GUARD_B - nold * BPS) & 15;                      // same for every lane ((Ts*BPS) % 16 == 0)
            v2f ph[M], dph[M], acc[M];
            v2f sw_ph[M], sw_dph[M];                       // wave-uniform: the new frame's oscillator at its first sample (the mixed block switches to it)
            const int n0 = TS * lb - nold + 1;             // recursion steps before this lane's first sample, counted from the frame's phase reference
            const int nold_blk = nold - TS * lb;           // samples of this block that are last frame's (<= 0: none)
            const bool oldl = nold_blk > 0;                // the block starts in last frame's samples: lanes 0, 1 (and 2 when nin = N - Ts/4)
            // Upstream keeps last frame's f_dc: samples mixed with LAST frame's tone estimates by an oscillator whose phase this frame's
            // continues. Both oscillators are at the phase reference (0) between the last old and the first new sample: an old position
            // n0 + k <= 0 steps before it is mixed with phasor((n0 + k) dtheta_old), at the gain the old recursion had reached there
            // (nin_prev + n0 + k steps after its renormalisation), which is what last frame's kernel stored in the f_dc ring of round 3.
#pragma unroll
            for (int m = 0; m < M; m++) {
                const float2 stn = a.t.osc_step[tix[m]], stp = a.t.osc_step[tixp[m]];          // uniform loads
                const float dn = a.t.osc_drift[tix[m]].x, dp = a.t.osc_drift[tixp[m]].x;
                const uint32_t th = (uint32_t)n0 * (oldl ? dthp[m] : dthv[m]);
                const float g = 1.0f + (oldl ? dp * (float)(ninp + n0) : dn * (float)n0);
                const float2 pcs = phasor(th);
                ph[m] = v2f{pcs.x * g, pcs.y * g};
                dph[m] = oldl ? v2f{stp.x, stp.y} : v2f{stn.x, stn.y};
                const float2 p1 = phasor(dthv[m]);
                const float g1 = 1.0f + dn;
                sw_ph[m] = v2f{p1.x * g1, p1.y * g1};
                sw_dph[m] = v2f{stn.x, stn.y};
                acc[m] = v2f{0.f, 0.f};
            }
            // without a last frame the old positions hold neutral samples (converted to exactly 0.0); csdr's u8 mapping has no such
            // byte value: that format zeroes the converted sample of an old position instead
            const int zero_blk = ninp ? 0 : nold_blk;
#pragma unroll
            for (int c = 0; c < TS / C::CHS; c++) {
                uint32_t rw[C::CH_DW];
                if (C::DIRECT && lb >= 3) {
                    // (blocks 3.. hold new samples only: from global memory, the frame's byte offset in the scalar operand)
#pragma unroll
                    for (int i = 0; i < C::CH_DW / 4; i++) {
                        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (TS * lb - nold) * BPS + c * C::CHS * BPS + 16 * i, (int)goff_frame, 0);
                        const uint4 u = __builtin_bit_cast(uint4, v);
                        rw[4 * i] = u.x; rw[4 * i + 1] = u.y; rw[4 * i + 2] = u.z; rw[4 * i + 3] = u.w;
                    }
                } else {
                    const unsigned char *bp = raw + boff + c * C::CHS * BPS;
                    if (al == 0) {
#pragma unroll
                        for (int i = 0; i < C::CH_DW / 4; i++) {
                            const uint4 v = ((const uint4 *)bp)[i];
                            rw[4 * i] = v.x; rw[4 * i + 1] = v.y; rw[4 * i + 2] = v.z; rw[4 * i + 3] = v.w;
                        }
                    } else if ((al & 7) == 0) {
#pragma unroll
                        for (int i = 0; i < C::CH_DW / 2; i++) { const uint2 v = ((const uint2 *)bp)[i]; rw[2 * i] = v.x; rw[2 * i + 1] = v.y; }
                    } else {
#pragma unroll
                        for (int i = 0; i < C::CH_DW; i++) rw[i] = ((const uint32_t *)bp)[i];
                    }
                }
#pragma unroll
                for (int kk = 0; kk < C::CHS; kk++) {
                    const int k = c * C::CHS + kk;
                    // the oscillator recursion is a serial chain; without this tie the optimiser converts all
                    // samples up front and holds far too many VGPRs
                    {
                        uint32_t &v0 = rw[BPS == 2 ? (kk >> 1) : (BPS == 4 ? kk : 2 * kk)];
                        if (M == 2) asm volatile("" : "+v"(v0), "+v"(ph[0]), "+v"(ph[M - 1]));
                        else asm volatile("" : "+v"(v0), "+v"(ph[0]), "+v"(ph[1]), "+v"(ph[M - 2]), "+v"(ph[M - 1]));
                    }
                    v2f x = decode<FMT>(rw, kk);
                    if (!InFmt<FMT>::NEUTRAL_OK && k < zero_blk) x = v2f{0.f, 0.f};   // old position of a stream's very first frame
                    // the one block that holds the last old and the first new sample (nold = 2 Ts -/+ Ts/4: block 1 at k = Ts - Ts/4,
                    // block 2 at k = Ts/4) changes oscillators there
                    if (k == Q || k == TS - Q) {
                        const bool sw = nold_blk == k;
#pragma unroll
                        for (int m = 0; m < M; m++) {
                            ph[m].x = sw ? sw_ph[m].x : ph[m].x; ph[m].y = sw ? sw_ph[m].y : ph[m].y;
                            dph[m].x = sw ? sw_dph[m].x : dph[m].x; dph[m].y = sw ? sw_dph[m].y : dph[m].y;
                        }
                    }
                    v2f nacc[M];
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        nacc[m] = mix_conj_acc(x, ph[m], acc[m]);
                        ph[m] = rot_step(ph[m], dph[m]);
                    }
                    // The new sums are pinned here: otherwise hipcc sinks every "acc += f + hv" to the end of the unrolled loop
                    // and keeps -- spills -- all Ts f and hv values until then. Tying the NEW value leaves the old one, which
                    // is the prefix sum at window start k, in its register without a copy.
                    if (M == 2) asm volatile("" : "+v"(nacc[0]), "+v"(nacc[M - 1]));
                    else asm volatile("" : "+v"(nacc[0]), "+v"(nacc[1]), "+v"(nacc[M - 2]), "+v"(nacc[M - 1]));
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        if (k % STEP == 0) fi[m][k / STEP] = acc[m];
                        acc[m] = nacc[m];
                    }
                    if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the unrolled loop's live set small
                }
            }
#pragma unroll
            for (int m = 0; m < M; m++) tot[m] = acc[m];
        }
        // every read of this frame's staged samples has been issued. The frame's last HIST raw samples become the next frame's old
        // positions: move them into the guard (whole dwords, source and destination disjoint), then request the next frame's
        // superset, which overwrites where they were (its start is known; its length only after this frame's timing estimate)
        PIRIP_PHASE_LANE(lane);
        if constexpr (C::DIRECT) {
            // (unstaged: the tail comes from global memory, copied into the guard by the same LDS-DMA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wave_lds_sync();
            const uint32_t toff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((pos + nin - HIST) * BPS));
#pragma unroll
            for (int i = 0; i < (C::TAIL_DW + kWave - 1) / kWave; i++)
                if (lane + i * kWave < C::TAIL_DW)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(raw + GUARD_B - HIST * BPS + i * 256), 4, lane * 4, toff + i * 256, 0, 0);
        } else {
            const uint32_t *tsrc = (const uint32_t *)(raw + GUARD_B + (nin - HIST) * BPS);
            uint32_t *tdst = (uint32_t *)(raw + GUARD_B - HIST * BPS);
            uint32_t tv[(C::TAIL_DW + kWave - 1) / kWave];
#pragma unroll
            for (int i = 0; i < (C::TAIL_DW + kWave - 1) / kWave; i++) tv[i] = tsrc[lane + i * kWave < C::TAIL_DW ? lane + i * kWave : 0];
#pragma unroll
            for (int i = 0; i < (C::TAIL_DW + kWave - 1) / kWave; i++) if (lane + i * kWave < C::TAIL_DW) tdst[lane + i * kWave] = tv[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_lds_sync();
        PIRIP_T_MARK(3);                                   // correlator
        dma_frame(pos + nin);
        ninp = nin;
#pragma unroll
        for (int m = 0; m < M; m++) { dthp[m] = dthv[m]; tixp[m] = tix[m]; }

        // ================= a-7: window sums (own suffix + next lane's prefix), |.|^2, fine-timing phasor sum ==========
        float tcr = 0.f, tci = 0.f;
        {
            v2f prpi{0.f, 0.f};
            v2f ft{0.f, 0.f};                                                // .x = sum over tones of |window sum|^2 (.y rides along unused)
#pragma unroll
            for (int q = 0; q < P; q++) {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const v2f own = fi[m][q];
                    v2f w0;                                                  // own suffix + next lane's prefix (DPP source operand)
                    w0.x = add_lane_up(own.x, tot[m].x - own.x);
                    w0.y = add_lane_up(own.y, tot[m].y - own.y);
                    fi[m][q] = w0;
                    // |w0|^2 (into / onto ft.x) as asm: left to hipcc, the SLP vectoriser pairs these across tones and pays five v_mov per window start
                    if (m == 0) asm("v_mul_f32 %0, %1, %1\n\tv_fma_f32 %0, %2, %2, %0" : "=&v"(ft.x) : "v"(w0.y), "v"(w0.x));
                    else { float t2; asm("v_mul_f32 %0, %2, %2\n\tv_fma_f32 %0, %3, %3, %0\n\tv_add_f32 %1, %1, %0" : "=&v"(t2), "+v"(ft.x) : "v"(w0.y), "v"(w0.x)); }
                }
                const v2f tp = *(const v2f *)&s_tph[q];    // exp(+j 2 pi q / P), uniform LDS read
                // (pr, pi) += ft1 * (cos, sin): one packed fma, ft1 broadcast from the low half (round 5: was two scalar fma)
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(prpi) : "v"(ft), "v"(tp));
                // one window start at a time: without this tie hipcc computes all P*M window sums first and spills
                asm volatile("" : "+v"(prpi), "+v"(fi[0][q]), "+v"(fi[M - 1][q]));
            }
            const float pr = prpi.x, pi = prpi.y;
            if (lane <= NSYM) {                            // (Nsym+1)*P window starts in all
                const float2 tgain = s_tgain[lane];
                tcr = pr * tgain.x - pi * tgain.y;
                tci = pr * tgain.y + pi * tgain.x;
            }
            tcr = wsum(tcr); tci = wsum(tci);
        }

        PIRIP_T_MARK(4);                                   // DMA issue, hist copy, window sums, timing reduction
        PIRIP_ARGS_REREAD();                               // (the output phase reads its pointers and strides afresh: nothing of them is live across the estimator and correlator)
        const int frame_bytes = d.pack_bits ? (d.Nbits + 7) / 8 : d.Nbits;
        const size_t orow = (size_t)(frame + out0);
        uint8_t *bits_o = a.io.bits ? a.io.bits + (size_t)sid * a.io.bits_stride + orow * frame_bytes : nullptr;
        float *filt_o = a.io.filt ? a.io.filt + (size_t)sid * a.io.filt_stride + orow * M * NSYM : nullptr;
        float *stats_o = a.io.stats ? a.io.stats + (size_t)sid * a.io.stats_stride + orow * PIRIP_STATS_PER_FRAME : nullptr;
        float f_est[kMaxTones] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < M; m++) {
            if constexpr (MASK) f_est[m] = (float)((bb - NDFT / 2) * d.Fs / NDFT) + (float)(m * d.tone_spacing);
            else f_est[m] = (float)freqi[m] * d.bin_hz;
        }

        const bool bad = isnan(tcr) || isnan(tci);
        int nin_next = nin;
        if (!bad) {
            // single precision (codec2 divides by 2 pi and smooths ppm in double): the results differ from the double path by
            // at most an ulp, far inside what the different summation order of the window sums already moves the estimate;
            // double-precision instructions in this once-per-frame block cost ~10 % of the kernel through register pressure
            const float norm_rx_timing = FMT == PIRIP_IN_CF32 ? atan2f(tci, tcr) * 0.15915494309189535f : atan2_turns(tci, tcr);
            const float rx_timing = norm_rx_timing * (float)P;
            const float d_norm = norm_rx_timing - sc_norm_rx_timing;
            sc_norm_rx_timing = norm_rx_timing;
            if (fabsf(d_norm) < 0.2f) {
                const float appm = (1e6f * d_norm) / (float)NSYM;      // (the expression every kernel and capture.hip's ppm recomputation share)
                sc_ppm = (0.9f * sc_ppm) + (0.1f * appm);
            }
            nin_next = N;
            if (!d.burst_mode) {
                if (norm_rx_timing > 0.25f) nin_next = N + Q;
                else if (norm_rx_timing < -0.25f) nin_next = N - Q;
            }

            // ================= a-8: resample, decide ====================================================================
            const int low_sample = __builtin_amdgcn_readfirstlane((int)floorf(rx_timing));
            const float fract = rx_timing - (float)low_sample;
            const int high_sample = __builtin_amdgcn_readfirstlane((int)ceilf(rx_timing));
            // f_int[(i+1)P + s]: s >= 0 -> lane i+1 register s, s < 0 -> lane i register P+s.
            // The register index is wave-uniform: a branch tree (the empty volatile asm keeps hipcc from folding the
            // cases back into a P-way v_cndmask chain per value) leaves one v_mov per selected register.
            v2f lo[M], hi[M];
            {
                const int ql = low_sample >= 0 ? low_sample : P + low_sample;
                const int qh = high_sample >= 0 ? high_sample : P + high_sample;
#define PIRIP_SEL_CASE(q) case q: if (q < P) { asm volatile(""); for (int m_ = 0; m_ < M; m_++) DST[m_] = fi[m_][q < P ? q : 0]; } break;
#define PIRIP_SELECT(idx) do { switch (idx) { \
    PIRIP_SEL_CASE(0) PIRIP_SEL_CASE(1) PIRIP_SEL_CASE(2) PIRIP_SEL_CASE(3) PIRIP_SEL_CASE(4) PIRIP_SEL_CASE(5) \
    PIRIP_SEL_CASE(6) PIRIP_SEL_CASE(7) PIRIP_SEL_CASE(8) PIRIP_SEL_CASE(9) PIRIP_SEL_CASE(10) PIRIP_SEL_CASE(11) \
    PIRIP_SEL_CASE(12) PIRIP_SEL_CASE(13) PIRIP_SEL_CASE(14) PIRIP_SEL_CASE(15) PIRIP_SEL_CASE(16) PIRIP_SEL_CASE(17) \
    PIRIP_SEL_CASE(18) PIRIP_SEL_CASE(19) PIRIP_SEL_CASE(20) PIRIP_SEL_CASE(21) PIRIP_SEL_CASE(22) PIRIP_SEL_CASE(23) \
    default: break; } } while (0)
                static_assert(P <= 24, "selection switch covers 24 window starts");
#pragma unroll
                for (int m = 0; m < M; m++) { lo[m] = fi[m][0]; hi[m] = fi[m][0]; }
#define DST lo
                PIRIP_SELECT(ql);
#undef DST
#define DST hi
                PIRIP_SELECT(qh);
#undef DST
#undef PIRIP_SELECT
#undef PIRIP_SEL_CASE
#pragma unroll
                for (int m = 0; m < M; m++) {
                    if (low_sample >= 0) { lo[m].x = lane_up(lo[m].x); lo[m].y = lane_up(lo[m].y); }
                    if (high_sample >= 0) { hi[m].x = lane_up(hi[m].x); hi[m].y = lane_up(hi[m].y); }
                }
            }
            float tmax[M];
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < M; m++) {
                v2f t;
                t.x = (1 - fract) * lo[m].x; t.y = (1 - fract) * lo[m].y;
                t.x = t.x + fract * hi[m].x; t.y = t.y + fract * hi[m].y;
                tmax[m] = (t.x * t.x) + (t.y * t.y);
                sum += tmax[m];
            }
            float mx = tmax[0]; int sym = 0;
#pragma unroll
            for (int m = 1; m < M; m++) if (tmax[m] > mx) { mx = tmax[m]; sym = m; }
            const bool act = lane < NSYM;
            if (bits_o && !d.pack_bits) {
                if (act) {
                    if (M == 2) bits_o[lane] = sym == 1;
                    else { bits_o[2 * lane + 1] = sym & 1; bits_o[2 * lane] = (sym & 2) >> 1; }
                }
            } else if (bits_o) {
                // 8 bits per byte, MSB first (codec2 freedv_pack order): wave ballots give the frame's bits as
                // 64-bit masks; lane j assembles byte j
                const unsigned long long mlo = __ballot(act && (sym & 1)), mhi = __ballot(act && (sym & 2));
                if (lane < frame_bytes) {
                    unsigned byte = 0;
                    if (M == 2) {
                        byte = __builtin_bitreverse32((unsigned)(mlo >> (8 * lane)) & 0xffu) >> 24;
                    } else {
                        const unsigned h4 = (unsigned)(mhi >> (4 * lane)) & 0xfu, l4 = (unsigned)(mlo >> (4 * lane)) & 0xfu;
#pragma unroll
                        for (int q = 0; q < 4; q++) byte |= (((h4 >> q) & 1u) << (7 - 2 * q)) | (((l4 >> q) & 1u) << (6 - 2 * q));
                    }
                    bits_o[lane] = (uint8_t)byte;
                }
            }
            if (act && filt_o) {
#pragma unroll
                for (int m = 0; m < M; m++) filt_o[m * NSYM + lane] = sqrtf(tmax[m]);
            }
            if constexpr (SOFT_OK) if (a.io.soft.llr) {
                // Bit LLRs from the soft magnitudes fsk_demod_sd would have handed over, computed exactly as the LLR stage computes
                // them from rx_filt (ldpc_kernels.hip: llr_tile_kernel; the checker (ldpc_oracle.c): oracle_ldpc_llr): per-symbol terms on
                // every lane, the frame's two sums by the wave reduction, ln I0 by table + linear interpolation, 4-FSK bits by max-log.
                // Square roots and the divisions by constants are IEEE operations: any correctly rounded form gives the same words. When every
                // |f|^2 of the frame is zero or >= 2^-96 (one wave-uniform test: always, short of denormal inputs) the roots are the estimator's
                // rsq form (6 instructions instead of sqrtf's 20) and x / 3, x / Nsym are x * (1/c) corrected once (3 instead of 11: exact
                // for every x >= 2^-125, tools/div_const_check.c; pirip_hip_selftest_div repeats it on the device).
                float mag[M], sum2 = 0.f, mx2 = 0.f;
                unsigned kmin = 0xffffffffu;
#pragma unroll
                for (int m = 0; m < M; m++) kmin = umin2(kmin, sqrt_key(tmax[m]));
                const bool quick = __all(!act || kmin >= 0x0f800000u - 1u);
                float ssig, snse;
                // (the receiver's defined summation order IS this kernel's wave reduction: ldpc_kernels.hip wave_order_sum)
                if (quick) {
#pragma unroll
                    for (int m = 0; m < M; m++) { mag[m] = sqrt_rn_normal<FMT != PIRIP_IN_CF32>(tmax[m]); const float p2 = mag[m] * mag[m]; sum2 = sum2 + p2; mx2 = p2 > mx2 ? p2 : mx2; }
                    ssig = wsum(act ? mx2 : 0.f); snse = wsum(act ? div_rn_const<M - 1>(sum2 - mx2) : 0.f);
                    ssig = div_rn_const<NSYM>(ssig);
                    snse = div_rn_const<NSYM>(snse) + 1e-12f;
                } else {
#pragma unroll
                    for (int m = 0; m < M; m++) { mag[m] = sqrtf(tmax[m]); const float p2 = mag[m] * mag[m]; sum2 = sum2 + p2; mx2 = p2 > mx2 ? p2 : mx2; }
                    ssig = wsum(act ? mx2 : 0.f); snse = wsum(act ? (sum2 - mx2) / (float)(M - 1) : 0.f);
                    ssig = ssig / (float)NSYM;
                    snse = (snse / (float)NSYM) + 1e-12f;
                }
                const int llr_map = a.io.soft.llr_map;                 // wave-uniform: codec2's mapping as recalled (default) or the exact Rician one
                const float g = llr_frame_gain_quick(llr_map, ssig, snse);
                float L[M];
                if (llr_map == kLlrRician) {
#pragma unroll
                    for (int m = 0; m < M; m++) L[m] = ln_i0_tab(s_lnI0, g * mag[m]);
                } else {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        const float x = g * mag[m];
                        int sg = x >= 1.0f ? 1 : 0;                          // (a chain of selects: 2 instructions per threshold, as a sum 3)
                        sg = x >= 2.0f ? 2 : sg; sg = x >= 5.0f ? 3 : sg; sg = x >= 20.0f ? 4 : sg;
                        const float4 cf = ((const float4 *)s_lnI0)[sg];
                        L[m] = (((cf.x * x) * x) + (cf.y * x)) + cf.z;       // = logbesseli0_upstream(x), operation for operation
                    }
                }
                float l0, l1 = 0.f;
                if (M == 2) l0 = L[0] - L[M - 1];
                else {
                    l0 = (L[0] > L[1] ? L[0] : L[1]) - (L[M - 2] > L[M - 1] ? L[M - 2] : L[M - 1]);      // MSB: symbols 0,1 vs 2,3
                    l1 = (L[0] > L[M - 2] ? L[0] : L[M - 2]) - (L[1] > L[M - 1] ? L[1] : L[M - 1]);      // LSB: symbols 0,2 vs 1,3
                }
                const float lmax = llr_map == kLlrRician ? kLlrMax : kLlrMaxUpstream;
                l0 = l0 > lmax ? lmax : (l0 < -lmax ? -lmax : l0);
                l1 = l1 > lmax ? lmax : (l1 < -lmax ? -lmax : l1);
                // soft bits are handed over as IEEE binary16 (round to nearest even); the hard decisions are the signs of THOSE values
                const _Float16 h0v = (_Float16)l0, h1v = (_Float16)l1;
                l0 = (float)h0v; l1 = (float)h1v;
                uint16_t *llr_o = a.io.soft.llr + (size_t)sid * a.io.soft.llr_stride + a.io.soft.bit0 + (size_t)frame * NBITS;
                if (act) {
                    if (M == 2) llr_o[lane] = __builtin_bit_cast(uint16_t, h0v);
                    else *(uint32_t *)(llr_o + 2 * lane) = (uint32_t)__builtin_bit_cast(uint16_t, h0v) | ((uint32_t)__builtin_bit_cast(uint16_t, h1v) << 16);
                }
                // hard decisions (llr < 0), 32 per word, first bit in the MSB, appended to the stream's bit string
                const unsigned long long h0 = __ballot(act && l0 < 0.0f), h1 = __ballot(act && l1 < 0.0f);
                if (M == 2) {
                    soft_append(__builtin_bitreverse32((uint32_t)h0), 32);
                    soft_append(__builtin_bitreverse32((uint32_t)(h0 >> 32)), NBITS - 32);
                } else {
#pragma unroll
                    for (int j = 0; j < (NBITS + 31) / 32; j++) {
                        const uint32_t z = spread16((uint32_t)(h0 >> (16 * j)) & 0xffffu) | (spread16((uint32_t)(h1 >> (16 * j)) & 0xffffu) << 1);
                        soft_append(__builtin_bitreverse32(z), (j + 1) * 32 <= NBITS ? 32 : NBITS - 32 * j);
                    }
                }
            }
            // SNRest / the smoothed EbNodB are per-frame outputs (stats) and stream state that only the LAST frame of a
            // call leaves behind: skip their wave reductions on frames where nobody can observe them
            const bool last_frame = (frame + 1 >= max_frames) || (pos + nin + nin_next > nsamp);
            if (stats_o || last_frame) {
                float sig = act ? mx : 0.f, nse = act ? (sum - mx) / (float)(M - 1) : 0.f;
                sig = wsum(sig); nse = wsum(nse) + 1e-12f;
                sig = sig / (float)NSYM; nse = nse / (float)NSYM;
                sc_SNRest = sig / nse;
                // codec2's other by-products (v_est, EbNodB and its smoothed form, MODEM_STATS.snr_est): kept in LDS between
                // observable frames so they cost no registers on the frames that skip this block
                float mean_e = wsum(act ? sqrtf(mx) : 0.f), std_e = wsum(act ? mx : 0.f);
                mean_e = mean_e / (float)NSYM;
                std_e = (std_e / (float)NSYM) - (mean_e * mean_e);
                // (single-precision throughout: (float)sqrt((double)x) == sqrtf(x) and .5*a + .5*b rounds the same either way;
                //  the 1e-6 guards and the divide differ from the double-precision original by < 1 ulp -- EbNodB is a logged
                //  figure, compared with a tolerance. Double-precision code here cost the whole kernel 10 % through register
                //  allocation although it runs once per call.)
                std_e = std_e > 0.0f ? sqrtf(std_e) : 0.0f;
                const float EbNodB = -6.0f + (20.0f * log10f((1e-6f + mean_e) / (1e-6f + std_e)));
                if (lane == 0) {
                    s_misc[wv][0] = (0.5f * s_misc[wv][0]) + (0.5f * EbNodB);      // snr_est
                    s_misc[wv][1] = EbNodB;
                    s_misc[wv][2] = sqrtf(sig - nse);                               // v_est
                    // rx_sig_pow / rx_nse_pow go straight to the stream's state and the stats row: three blocks of this kernel fill a
                    // CU's LDS to within 16 bytes, there is no room for two more floats per stream
                    a.s.scal[sid].rx_sig_pow = sig; a.s.scal[sid].rx_nse_pow = nse;
                    if (stats_o) { stats_o[8] = sig; stats_o[9] = nse; }
                }
            }
        } else {
            for (int i = lane; i < frame_bytes; i += kWave) if (bits_o) bits_o[i] = 0;
            for (int i = lane; i < M * NSYM; i += kWave) if (filt_o) filt_o[i] = 0.f;
            if (stats_o && lane == 0) { stats_o[8] = 0.f; stats_o[9] = 0.f; }
            if constexpr (SOFT_OK) if (a.io.soft.llr) {            // zero magnitudes map to zero LLRs / zero hard bits
                uint16_t *llr_o = a.io.soft.llr + (size_t)sid * a.io.soft.llr_stride + a.io.soft.bit0 + (size_t)frame * NBITS;
                for (int i = lane; i < NBITS; i += kWave) llr_o[i] = 0;
                for (int j = 0; j < (NBITS + 31) / 32; j++) soft_append(0u, (j + 1) * 32 <= NBITS ? 32 : NBITS - 32 * j);
            }
        }
        if constexpr (MASK) last_freqi0 = bb;
        else { last_freqi0 = freqi[0]; last_freqi1 = freqi[M > 1 ? 1 : 0]; last_freqi2 = freqi[M > 2 ? 2 : 0]; last_freqi3 = freqi[M > 3 ? 3 : 0]; }
        if (stats_o && lane == 0) {
            stats_o[0] = f_est[0]; stats_o[1] = f_est[1]; stats_o[2] = f_est[2]; stats_o[3] = f_est[3];
            stats_o[4] = sc_norm_rx_timing; stats_o[5] = sc_SNRest; stats_o[6] = (float)nin_next; stats_o[7] = sc_ppm;
        }
        PIRIP_T_MARK(5);                                   // atan2, decisions, outputs
        pos += nin;
        nin = __builtin_amdgcn_readfirstlane(nin_next);
        frame++;
        wave_lds_sync();
    }

    // ---- save stream state ---------------------------------------------------------------------------------------------
    if constexpr (SOFT_OK) if (words_o && soft_cb > 0 && lane0 == 0) words_o[soft_w] = soft_carry;   // the last, partly filled word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the speculative next-frame DMA must not outlive the LDS allocation
#ifdef PIRIP_WAVE_TIMING
    if (a.io.stats && lane0 == 0 && sid == nstreams / 2)
        for (int i = 0; i < 8; i++) a.io.stats[(size_t)sid * a.io.stats_stride + i] = (float)t_acc_[i];
#endif
#pragma unroll
    for (int b = 0; b < NB; b++) if (!BAND || lane0 < 16) a.s.Sf[(size_t)sid * NDFT + own_sfi(lane0, b)] = Sf[b];
    wave_lds_sync();
    for (int i = lane0; i < GUARD_B / 4; i += kWave) st32[i] = ((const uint32_t *)raw)[i];     // the raw tail, as the guard holds it
    const int lane = lane0;
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) { st32[C::TRAILER_DW + m] = dthp[m]; st32[C::TRAILER_DW + M + m] = (uint32_t)tixp[m]; }
        st32[C::TRAILER_DW + 2 * M] = (uint32_t)ninp;
        StreamScalars sc = a.s.scal[sid];
        sc.nin = nin; sc.norm_rx_timing = sc_norm_rx_timing; sc.ppm = sc_ppm; sc.SNRest = sc_SNRest;
        sc.snr_est = s_misc[wv][0]; sc.EbNodB = s_misc[wv][1]; sc.v_est = s_misc[wv][2];
        if (frame > frame_first) {
            const int fq[4] = {last_freqi0, last_freqi1, last_freqi2, last_freqi3};
            for (int m = 0; m < kMaxTones; m++) {
                if (MASK) sc.f_est[m] = m < M ? (float)((fq[0] - NDFT / 2) * d.Fs / NDFT) + (float)(m * d.tone_spacing) : 0.f;
                else sc.f_est[m] = m < M ? (float)fq[m] * d.bin_hz : 0.f;
            }
        }
        a.s.scal[sid] = sc;
        if (a.io.nframes) a.io.nframes[sid] = (int32_t)frame;
        if (a.io.consumed) a.io.consumed[sid] = pos;
    }
#undef a
#undef d
#undef PIRIP_ARGS_REREAD
}

// ---- self-test of the estimator's square root (measurement behind the FINITE variant's comment) ----------------------------------
namespace {
__global__ void sqrt_selftest_kernel(unsigned long long *bad, unsigned lo, unsigned hi)
{
    unsigned long long c = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        const float want = (float)sqrt((double)x);             // double rounding is innocuous for sqrt
        if (__builtin_bit_cast(unsigned, sqrt_rn_normal<true>(x)) != __builtin_bit_cast(unsigned, want)) c++;
        if (x != 0.0f && __builtin_bit_cast(unsigned, sqrt_rn_normal<true, true>(x)) != __builtin_bit_cast(unsigned, want)) c++;      // (the unguarded form, where it runs)
        if (__builtin_bit_cast(unsigned, sqrt_rn_normal<false>(x)) != __builtin_bit_cast(unsigned, want)) c += 1ull << 32;
    }
    if (c) atomicAdd(bad, c);
}
}  // namespace

hipError_t selftest_sqrt(unsigned long long *mismatches)
{
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(*d));
    if (e != hipSuccess) return e;
    e = hipMemset(d, 0, sizeof(*d));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(4096), dim3(256), 0, 0, d, 0u, 0u);                    // x = 0
        hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(4096), dim3(256), 0, 0, d, 0x0f800000u, 0x7f7fffffu);  // [2^-96, FLT_MAX]
        e = hipMemcpy(mismatches, d, sizeof(*d), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipGetLastError();
    }
    (void)hipFree(d);
    return e;
}

// ---- self-test of the fused hand-over's divisions by constants (div_rn_const) against the device's IEEE quotient -----------------------
namespace {
template <int C>
__global__ void div_selftest_kernel(unsigned long long *bad, unsigned lo, unsigned hi)
{
    unsigned long long c = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __builtin_bit_cast(float, (unsigned)b);
        if (__builtin_bit_cast(unsigned, div_rn_const<C>(x)) != __builtin_bit_cast(unsigned, x / (float)C)) c++;
    }
    if (c) atomicAdd(bad, c);
}
}  // namespace

hipError_t selftest_div(unsigned long long *mismatches)
{
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc(&d, 2 * sizeof(*d));
    if (e != hipSuccess) return e;
    e = hipMemset(d, 0, 2 * sizeof(*d));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(div_selftest_kernel<3>, dim3(4096), dim3(256), 0, 0, d, 0u, 0u);                          // x = 0
        hipLaunchKernelGGL(div_selftest_kernel<3>, dim3(4096), dim3(256), 0, 0, d, 0x01000000u, 0x7f7fffffu);        // [2^-125, FLT_MAX]
        hipLaunchKernelGGL(div_selftest_kernel<50>, dim3(4096), dim3(256), 0, 0, d + 1, 0u, 0u);
        hipLaunchKernelGGL(div_selftest_kernel<50>, dim3(4096), dim3(256), 0, 0, d + 1, 0x01000000u, 0x7f7fffffu);
        unsigned long long h[2] = {0, 0};
        e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipGetLastError();
        const unsigned long long a = h[0] > 0xffffffffull ? 0xffffffffull : h[0], b = h[1] > 0xffffffffull ? 0xffffffffull : h[1];
        *mismatches = (b << 32) | a;
    }
    (void)hipFree(d);
    return e;
}

// ---- instances and dispatch ----------------------------------------------------------------------------------------------
namespace {

struct WaveInst {
    int M, Ts, P, Nsym, Ndft, fmt, fft_fma, mask, wpb, wps, band;
    hipError_t (*launch)(const DemodArgs &, int, hipStream_t);
};

template <int M, int TS, int P, int NSYM, int NDFT, int FMT, int WPB, int WPS, bool FMA, bool MASK, int BAND = 0>
hipError_t launch_inst(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const dim3 g((nstreams + WPB - 1) / WPB), b(kWave * WPB);
    const size_t dyn = a.io.soft.llr ? 258 * sizeof(float) : 0;          // the ln I0 table of the fused FSK_LDPC hand-over
#ifdef PIRIP_WAVE_TIMING
    {
        const char *e = getenv("PIRIP_WAVE_STOP");
        const int stop = e ? atoi(e) : -1;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wave_stop_phase), &stop, sizeof(int), 0, hipMemcpyHostToDevice, stream);
        (void)hipStreamSynchronize(stream);
    }
#endif
    hipLaunchKernelGGL((fsk_demod_wave_kernel<M, TS, P, NSYM, NDFT, FMT, WPB, WPS, FMA, MASK, BAND>), g, b, dyn, stream, a, nstreams);
    return hipGetLastError();
}

#define PIRIP_WAVE_INST(M, TS, P, NDFT, FMT, WPB, WPS) {M, TS, P, 50, NDFT, FMT, 0, 0, WPB, WPS, 0, launch_inst<M, TS, P, 50, NDFT, FMT, WPB, WPS, false, false>}
#define PIRIP_WAVE_INST_FMA(M, TS, P, NDFT, FMT, WPB, WPS) {M, TS, P, 50, NDFT, FMT, 1, 0, WPB, WPS, 0, launch_inst<M, TS, P, 50, NDFT, FMT, WPB, WPS, true, false>}
#define PIRIP_WAVE_INST_MASK(M, TS, P, NDFT, FMT, WPB, WPS) {M, TS, P, 50, NDFT, FMT, 0, 1, WPB, WPS, 0, launch_inst<M, TS, P, 50, NDFT, FMT, WPB, WPS, false, true>}
#define PIRIP_WAVE_INST_BAND(M, TS, P, NDFT, FMT, WPB, WPS, B) {M, TS, P, 50, NDFT, FMT, 0, 0, WPB, WPS, B, launch_inst<M, TS, P, 50, NDFT, FMT, WPB, WPS, false, false, B>}
#ifndef PIRIP_N128_WPB          // (build-time experiment knobs for the Ndft = 128 2-FSK instances: streams per block, waves per SIMD)
// complex-float instances: unstaged (PIRIP_F32_DIRECT) they are no longer bound by LDS: four streams per block, as many waves as the registers allow
#if PIRIP_F32_DIRECT
#define PIRIP_F32_WPB 4
#define PIRIP_F32_WPS(M) ((M) == 2 ? 3 : 2)      /* Ts = 40; 4-FSK at three waves per SIMD spills 3-9 registers */
#define PIRIP_F32_WPS256(M) 3                     /* Ts = 18, 20: every instance fits 168 VGPR */
#else
#define PIRIP_F32_WPB 2
#define PIRIP_F32_WPS(M) 1
#define PIRIP_F32_WPS256(M) 1
#endif
#if PIRIP_F32_DIRECT && PIRIP_S16_DIRECT
#define PIRIP_S16_WPS(M) ((M) == 2 ? 3 : 2)
#else
#define PIRIP_S16_WPS(M) 2
#endif
#define PIRIP_N128_WPB 4
#define PIRIP_N128_WPS 4
#endif
const WaveInst kInst[] = {
#ifdef PIRIP_WAVE_PROBE      // compile-time experiments: one instance only
    PIRIP_WAVE_INST(2, 24, PIRIP_WAVE_PROBE_P, 256, PIRIP_IN_CU8_FSKDEMOD, 4, PIRIP_WAVE_PROBE),
#else
    // Ts = 24 (Fs 240k / Rs 10k), both 8-bit front ends. P = 24: `fsk_demod -p 24` (README.md:105); P = 8: fsk_demod's default;
    // P = 6: what rtl_fsk derives from Ts = 24. _MASK: the `--mask` comb estimator (README.md:239-297)
    PIRIP_WAVE_INST(2, 24, 24, 256, PIRIP_IN_CU8_FSKDEMOD, 4, 3),
    PIRIP_WAVE_INST_FMA(2, 24, 24, 256, PIRIP_IN_CU8_FSKDEMOD, 4, 3),     // opt-in fused complex multiply (headline shape only)
    PIRIP_WAVE_INST(2, 24, 24, 256, PIRIP_IN_CU8_CSDR, 4, 3),
    // opt-in band-only estimator (pirip_hip_set_estimator_band_only): the `fsk_demod -p 24` shape, both 8-bit front ends
    PIRIP_WAVE_INST_BAND(2, 24, 24, 256, PIRIP_IN_CU8_FSKDEMOD, 4, 3, 2), PIRIP_WAVE_INST_BAND(2, 24, 24, 256, PIRIP_IN_CU8_CSDR, 4, 3, 2),
    // ... and the 4-FSK P = 8 shape with a search range inside bins 0 .. 63 (BASELINE configs[3] as bench_configs.py sets it up: 500 .. 60000 Hz)
    PIRIP_WAVE_INST_BAND(4, 24, 8, 256, PIRIP_IN_CU8_FSKDEMOD, 4, PIRIP_M4_WPS, 4), PIRIP_WAVE_INST_BAND(4, 24, 8, 256, PIRIP_IN_CU8_CSDR, 4, PIRIP_M4_WPS, 4),
#define PIRIP_TS24(M, P, WPB, WPS) \
    PIRIP_WAVE_INST(M, 24, P, 256, PIRIP_IN_CU8_FSKDEMOD, WPB, WPS), PIRIP_WAVE_INST(M, 24, P, 256, PIRIP_IN_CU8_CSDR, WPB, WPS), \
    PIRIP_WAVE_INST_MASK(M, 24, P, 256, PIRIP_IN_CU8_FSKDEMOD, WPB, WPS), PIRIP_WAVE_INST_MASK(M, 24, P, 256, PIRIP_IN_CU8_CSDR, WPB, WPS)
    PIRIP_TS24(2, 8, 4, 3), PIRIP_TS24(2, 6, 4, 3), PIRIP_TS24(4, 8, 4, PIRIP_M4_WPS), PIRIP_TS24(4, 6, 4, PIRIP_M4_WPS),    // (4-FSK at 9, 10 or 11 waves per CU -- blocks of 3, 5 or 11 streams, 168 VGPR -- measured slower per stream than 8: two waves already keep a SIMD's issue port busy)
#undef PIRIP_TS24
    // Ts = 40 (Fs 40k / Rs 1k): s16 behind the csdr decimator (README.md:109), f32 inside rtl_fsk (-a 40000 -r 1000: script/ping:47,
    // script/frame_repeater:36; 4-FSK with --mask: README.md:239). P = 8: fsk_demod's default, P = 10: rtl_fsk's.
    // (s16: 5 waves per block, 10 per CU, measured slower than 4 / 8: uneven SIMD load)
#define PIRIP_TS40(M, P) \
    PIRIP_WAVE_INST(M, 40, P, 512, PIRIP_IN_CS16, 4, PIRIP_S16_WPS(M)), PIRIP_WAVE_INST(M, 40, P, 512, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS(M)), \
    PIRIP_WAVE_INST_MASK(M, 40, P, 512, PIRIP_IN_CS16, 4, PIRIP_S16_WPS(M)), PIRIP_WAVE_INST_MASK(M, 40, P, 512, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS(M))
    PIRIP_TS40(2, 8), PIRIP_TS40(2, 10), PIRIP_TS40(4, 8), PIRIP_TS40(4, 10),
#undef PIRIP_TS40
    // Ts = 20 (rtl_fsk -a 200000 -r 10000 [-m 4] [--mask 10000]: README.md:262,292,297), float samples from the in-process decimator;
    // 6 FFTs of 256 per frame = one full batch of four and a half-empty one
    PIRIP_WAVE_INST(2, 20, 10, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(2)), PIRIP_WAVE_INST_MASK(2, 20, 10, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(2)),
    PIRIP_WAVE_INST(4, 20, 10, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(4)), PIRIP_WAVE_INST_MASK(4, 20, 10, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(4)),
    // Ts = 18 (rtl_fsk -a 180000 -r 10000 -m 4 --mask 10000: README.md:286); nin moves in steps of Ts/4 = 4 samples
    PIRIP_WAVE_INST(2, 18, 9, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(2)), PIRIP_WAVE_INST_MASK(2, 18, 9, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(2)),
    PIRIP_WAVE_INST(4, 18, 9, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(4)), PIRIP_WAVE_INST_MASK(4, 18, 9, 256, PIRIP_IN_CF32, PIRIP_F32_WPB, PIRIP_F32_WPS256(4)),
    // Ts = 10 / Ndft = 128 (rtl_fsk -a 100000 -r 10000: README.md:196) and Ts = 8 / Ndft = 128 (rtl_fsk -s 2400000 -a 80000 -r 10000 on a
    // Pi: README.md:172), float samples from the in-process decimator; all of a frame's 6 / 5 FFTs in one batch of eight
    PIRIP_WAVE_INST(2, 10, 10, 128, PIRIP_IN_CF32, PIRIP_N128_WPB, PIRIP_N128_WPS), PIRIP_WAVE_INST_MASK(2, 10, 10, 128, PIRIP_IN_CF32, PIRIP_N128_WPB, PIRIP_N128_WPS),
    PIRIP_WAVE_INST(4, 10, 10, 128, PIRIP_IN_CF32, 2, 2), PIRIP_WAVE_INST_MASK(4, 10, 10, 128, PIRIP_IN_CF32, 2, 2),
    PIRIP_WAVE_INST(2, 8, 8, 128, PIRIP_IN_CF32, PIRIP_N128_WPB, PIRIP_N128_WPS), PIRIP_WAVE_INST_MASK(2, 8, 8, 128, PIRIP_IN_CF32, PIRIP_N128_WPB, PIRIP_N128_WPS),
    PIRIP_WAVE_INST(4, 8, 8, 128, PIRIP_IN_CF32, 2, 2), PIRIP_WAVE_INST_MASK(4, 8, 8, 128, PIRIP_IN_CF32, 2, 2),
#endif
};
#undef PIRIP_WAVE_INST
#undef PIRIP_WAVE_INST_FMA
#undef PIRIP_WAVE_INST_MASK
#undef PIRIP_WAVE_INST_BAND

const WaveInst *find_inst(const FskDims &d)
{
    if (!d.recalled_fast_ok) return nullptr;            // (the instances carry the recalled constants' default values: pirip_fsk_recalled)
    for (const WaveInst &w : kInst)
        if (w.M == d.M && w.Ts == d.Ts && w.P == d.P && w.Nsym == d.Nsym && w.Ndft == d.Ndft && w.fmt == d.in_format && w.fft_fma == d.fft_fma &&
            w.mask == (d.freq_est_type != 0) && w.band == d.est_band) return &w;
    return nullptr;
}

}  // namespace

bool demod_wave_applicable(const FskDims &d) { return find_inst(d) != nullptr; }

// streams of this configuration's instance that one CU holds at a time (one wave each: waves per SIMD x 4 SIMDs; 0: no instance)
int demod_wave_streams_per_cu(const FskDims &d) { const WaveInst *w = find_inst(d); return w ? 4 * w->wps : 0; }

bool demod_wave_soft_capable(const FskDims &d) { return find_inst(d) != nullptr && d.P <= 10 && d.Nsym == 50; }

int demod_wave_describe(const FskDims &d, char *buf, size_t n)
{
    const WaveInst *w = find_inst(d);
    if (!w) return 0;
    static const char *const fmt_name[] = {"u8 -d", "u8 csdr", "s16", "f32"};
    return snprintf(buf, n, "fsk_demod_wave_kernel<M=%d,Ts=%d,P=%d,Nsym=%d,Ndft=%d,%s,%s%s%d streams/block,%d waves/SIMD>", w->M, w->Ts, w->P, w->Nsym,
                    w->Ndft, fmt_name[w->fmt & 3], w->mask ? "mask estimator," : w->band ? "band-only estimator," : "", w->fft_fma ? "fused FFT multiply," : "", w->wpb, w->wps);
}

int64_t demod_wave_max_samples(const FskDims &d)
{
    const int bps = d.in_format == PIRIP_IN_CF32 ? 8 : d.in_format == PIRIP_IN_CS16 ? 4 : 2;
    return 0x7fffff00LL / bps * 2 / 2;                      // 32-bit buffer-descriptor range in bytes
}

hipError_t launch_demod_wave(const DemodArgs &a, int nstreams, hipStream_t stream)
{
    const WaveInst *w = find_inst(a.d);
    if (!w || a.io.nsamp > demod_wave_max_samples(a.d)) return hipErrorNotSupported;
    return w->launch(a, nstreams, stream);
}

}  // namespace pirip
