// pirip_amd/csrc/fsk_plan.hpp -- host-side plan of one FSK demodulator configuration.
//
// Derives every constant codec2's fsk_create_core() keeps in struct FSK and builds the
// read-only tables the HIP kernels consume (Hann window, FFT twiddles / stage list / leaf
// permutation, u8 conversion LUT, fine-timing phasors, mask-estimator tooth positions).
// [UPSTREAM-RECALLED codec2 src/fsk.c: fsk_create_core, fsk_generate_hann_table,
//  fsk_demod_freq_est; src/kiss_fft.c: kiss_fft_alloc, kf_factor. Reference call sites that
//  fix the parameters: /root/reference/README.md:105,109, test/loopback_rtl_sdr.sh:16.]
//
// Everything here is plain host C++ (no HIP) so the CPU-only tools can use it too.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pirip_hip.h"   // pirip_fsk_recalled: the recalled constants as data (defaults = today's values)

namespace pirip {

constexpr int kMaxTones = 4;
constexpr int kMaxStages = 16;

struct FftStage { int radix; int m; int fstride; };   // executed leaf (m small) first

// Plain-old-data block handed to the kernels by value (lives in SGPRs / kernarg).
struct FskDims {
    int Fs, Rs, M, P, Nsym;
    int Ts, N, Nmem, Ndft, Nbits, nint;      // nint = (Nsym+1)*P
    int est_st, est_en, f_zero;              // peak search: bins [st,en), blank +-f_zero
    int freq_est_type;                       // 0 peak, 1 mask
    int tone_spacing;
    int mask_len;                            // len_mask of the comb
    int n_teeth;                             // number of 1-entries of the comb
    int in_format;
    int hist_len;                            // 2*Ts + Ts/4 : integrator memory kept between frames
    int grp;                                 // general kernel: samples per stored integrator-memory entry: gcd(Ts/P, Ts/4, hist_len)
                                             // -- Ts/P when every frame shift is a whole number of window steps
    int nstages;
    int pack_bits;                           // 0: one byte per bit (fsk_demod's stdout format); 1: 8 bits per byte, MSB first
    int burst_mode;                          // fsk_enable_burst_mode(): nin stays N (no timing-driven resizing)
    int fft_fma;                             // opt-in (PIRIP_FFT_FMA=1): fused complex multiply in the estimator FFT where an instance exists
    int est_band;                            // opt-in (pirip_hip_set_estimator_band_only): 0 = full estimator; 2 / 4 = Sf maintained only for FFT bins 0 .. 31 /
                                             // 0 .. 63 of Ndft = 256 (a band that holds the peak search's range); every output is unchanged
    float tc, one_minus_tc;
    float bin_hz;                            // (float)Fs/(float)Ndft
    // the recalled constants a kernel reads (pirip_fsk_recalled; the specialised instances are built around the defaults and are
    // only chosen when recalled_fast_ok): nin moves by nin_step samples beyond |norm_rx_timing| > nin_thresh; int16 samples are
    // divided by s16_scale; u8_table: the u8 map comes from the plan's 256-entry table; sf_power 1: |X|^2 is smoothed into Sf
    int nin_step;
    float nin_thresh;
    float s16_scale;
    int u8_table;
    int sf_power;
    int recalled_fast_ok;
    int block_stagger;                       // block kernel: start-up stagger of co-resident workgroups, in s_sleep(127) units per wave slot (0: none)
};

struct FskPlan {
    FskDims d{};
    FftStage stages[kMaxStages]{};
    std::vector<float> hann;        // [Ndft]
    std::vector<float> twiddle;     // [Ndft][2] (cos, sin) of -2*pi*i/Ndft, (float) of double
    std::vector<uint16_t> leaf_perm;// [Ndft] input index read by FFT work-array slot n
    std::vector<uint16_t> leaf_iperm;// [Ndft] inverse: FFT work-array slot fed by input index i
    std::vector<float> u8_lut;      // [256] conversion of the configured u8 format
    std::vector<float> timing_ph;   // [P][2] exp(+j*2*pi*k/P)
    std::vector<int16_t> teeth;     // [n_teeth] comb tooth offsets, ascending
    std::vector<uint32_t> mask_dtheta; // mask method: [Ndft*M] per-sample phase step (2^32 = one turn) of the upstream oscillator for
                                       // comb position b, tone m (index b*M + m); peak method: unused ([kMaxTones] zeros)
    // Drift model of the upstream recursive oscillators (see DESIGN.md "tracking the recursion"):
    // codec2 advances phi_c by a float32-rounded (cosf,sinf) pair once per sample, so |phi_c| and
    // its phase drift linearly inside a frame: after n steps gain ~ 1 + a*n, phase error ~ d*n.
    // osc_drift[b] = (a, d) for the tone whose estimate is table entry b:
    //   peak method: b = bin index in [0,Ndft) (freqi + Ndft/2);  mask method: b = bmax*M + m.
    std::vector<float> osc_drift;      // [Ndft*(mask?M:1)][2]
    std::vector<float> osc_step;       // same indexing, (cosf(w), sinf(w)): the rounded per-sample multiplier
    // Fine-timing phasor exactly as the upstream recursion produces it: phi_ft[0]=1, phi_ft[i+1]=phi_ft[i]*dphift
    std::vector<float> timing_rec;     // [nint][2]
    // fast kernel (Ndft == 256): per 16-lane-group lane e: hann16[16] | tw3[3][2] | tw4[4][3][2] | pad -> 48 floats
    std::vector<float> fast_tab;       // [16][48]; Ndft == 512: the wave kernel's tables (layout in fsk_plan.cpp)
    float tw_s2[18];                   // stage-2 twiddles tw[16k*r], k=1..3, r=1..3, (re,im)
    // returns 0 on success, <0 if codec2 would have asserted
    int init(int Fs, int Rs, int M, int P, int Nsym, int est_min, int est_max,
             int freq_est_type, int tone_spacing, int in_format, const pirip_fsk_recalled *recalled = nullptr);
};

// Tx-side / measurement-instrument helpers shared by the CPU tools (fsk_mod,
// fsk_get_test_bits, fsk_put_test_bits) -- plain scalar C++, never on the GPU path.
struct FskMod {
    int Fs, Rs, M, Ts, f1_tx, tone_spacing;
    float ph_re = 1.0f, ph_im = 0.0f;        // tx_phase_c
    void init(int Fs_, int Rs_, int M_, int f1, int spacing);
    // nbits input bits (one per byte) -> nsym*Ts samples; complex interleaved or real
    void mod(const uint8_t *bits, int nbits, float *out, bool complex_out);
};

void test_frame_bits(uint8_t *frame, int framesize);     // glibc srand(158324)/rand()&1

struct PutBits {
    int framesize; float valid_thresh;
    std::vector<uint8_t> tx, rx;
    long bitcnt = 0, biterr = 0; int packetcnt = 0;
    void init(int framesize_, float valid_thresh_);
    // returns true when this bit completed a valid packet (errs reported through *errs_out)
    bool push(uint8_t bit, int *errs_out);
    float ber() const { return bitcnt ? (float)biterr / (float)bitcnt : 0.5f; }
};

// csdr filter design (host) [UPSTREAM-RECALLED csdr libcsdr.c: firdes_filter_len,
// firdes_lowpass_f, firdes_wkernel_hamming]
int csdr_filter_len(float transition_bw);
void csdr_lowpass_hamming(float *taps, int length, float cutoff_rate);
void csdr_lowpass(float *taps, int length, float cutoff_rate, int window);   // 0 boxcar, 1 Blackman, 2 Hamming (csdr window_t)
// peak-search range of fsk_set_freq_est_limits(): fills est_st / est_en, or returns false where codec2 asserts
bool fsk_est_range(int Fs, int Ndft, int est_min, int est_max, int *st, int *en);
void recalled_defaults(pirip_fsk_recalled *r);                     // today's values (pirip_hip_recalled_defaults)
bool recalled_from_env(pirip_fsk_recalled *r);                     // PIRIP_RECALLED="field=value,..." over what r holds; false: unknown field

}  // namespace pirip
