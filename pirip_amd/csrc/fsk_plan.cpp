// pirip_amd/csrc/fsk_plan.cpp -- see fsk_plan.hpp. Host-only; build with -ffp-contract=off so
// the tables are the float32 values a baseline x86-64 build of the upstream C computes.
#include "fsk_plan.hpp"

#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/pirip_hip.h"

namespace pirip {

namespace {
struct cf { float re, im; };
inline cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
}  // namespace

int FskPlan::init(int Fs, int Rs, int M, int P, int Nsym, int est_min, int est_max,
                  int freq_est_type, int tone_spacing, int in_format, const pirip_fsk_recalled *recalled)
{
    pirip_fsk_recalled rc;
    recalled_defaults(&rc);
    const pirip_fsk_recalled dflt = rc;
    if (recalled) rc = *recalled;
    if (!(rc.tc > 0.0f && rc.tc <= 1.0f) || !(rc.est_space_rs >= 0.0f) || !(rc.nin_threshold > 0.0f && rc.nin_threshold < 0.5f) || rc.nin_step_div < 2 ||
        !(rc.s16_scale > 0.0f) || !(rc.u8d_scale > 0.0f) || rc.ndft_rule < 0 || rc.ndft_rule > 1 || rc.sf_power < 0 || rc.sf_power > 1 ||
        rc.hann_denominator_ndft < 0 || rc.hann_denominator_ndft > 1)
        return PIRIP_ERR_BAD_CONFIG;
    // the conditions fsk_create_core() asserts on
    if (Fs <= 0 || Rs <= 0 || P <= 0 || Nsym <= 0) return PIRIP_ERR_BAD_CONFIG;
    if (Fs % Rs) return PIRIP_ERR_BAD_CONFIG;
    if ((Fs / Rs) % P) return PIRIP_ERR_BAD_CONFIG;
    if (P < 4) return PIRIP_ERR_BAD_CONFIG;
    if (M != 2 && M != 4) return PIRIP_ERR_BAD_CONFIG;
    if (in_format < PIRIP_IN_CU8_FSKDEMOD || in_format > PIRIP_IN_CF32) return PIRIP_ERR_BAD_CONFIG;

    // frequency-estimator FFT size: bins within 10 % of the symbol rate, next power of two [recalled; ndft_rule 1: the older rule,
    // the largest power of two that fits a frame]
    float bin_width_Hz = 0.1 * Rs;
    float Ndft_f = (float)Fs / bin_width_Hz;
    Ndft_f = std::pow(2.0, std::ceil(std::log2((double)Ndft_f)));
    int Ndft = (int)Ndft_f;
    if (rc.ndft_rule == 1) { Ndft = 1; while (2 * Ndft <= (Fs / Rs) * Nsym) Ndft *= 2; }
    if (Ndft < 8 || Ndft > 16384) return PIRIP_ERR_BAD_CONFIG;
    if ((Fs / Rs) / rc.nin_step_div < 1) return PIRIP_ERR_BAD_CONFIG;

    d.Fs = Fs; d.Rs = Rs; d.M = M; d.P = P; d.Nsym = Nsym;
    d.Ts = Fs / Rs;
    d.N = d.Ts * Nsym;
    d.Nmem = d.N + 2 * d.Ts;
    d.Ndft = Ndft;
    d.Nbits = (M == 2) ? Nsym : 2 * Nsym;
    d.nint = (Nsym + 1) * P;
    d.tc = rc.tc;
    d.one_minus_tc = 1 - d.tc;
    d.nin_step = d.Ts / rc.nin_step_div;
    d.nin_thresh = rc.nin_threshold;
    d.s16_scale = rc.s16_scale;
    d.sf_power = rc.sf_power;
    d.u8_table = (in_format == PIRIP_IN_CU8_FSKDEMOD && (rc.u8d_offset != dflt.u8d_offset || rc.u8d_scale != dflt.u8d_scale)) ? 1 : 0;
    // the fields the specialised kernels are built around
    d.recalled_fast_ok = (rc.nin_threshold == dflt.nin_threshold && rc.nin_step_div == dflt.nin_step_div && rc.s16_scale == dflt.s16_scale &&
                          !d.u8_table && rc.ndft_rule == dflt.ndft_rule && rc.sf_power == dflt.sf_power) ? 1 : 0;
    d.freq_est_type = freq_est_type ? 1 : 0;
    d.tone_spacing = tone_spacing;
    d.in_format = in_format;
    d.hist_len = 2 * d.Ts + d.nin_step;
    {
        // nin moves by Ts/4 samples: stored groups stay aligned when the group size divides the window step, that shift and
        // the kept tail (Ts = 240, P = 15 -- rtl_fsk -r 1000 at 240 kS/s, README.md:152,184,239: step 16, shift 60 -> groups of
        // 4; with single samples that configuration's integrator memory alone is 100 KB of LDS and does not fit)
        const int step = d.Ts / P;
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        d.grp = gcd(gcd(step, d.nin_step > 0 ? d.nin_step : step), d.hist_len);
        if (d.grp < 1) d.grp = 1;
    }
    d.burst_mode = 0;
    d.block_stagger = 0;
    d.fft_fma = 0;
    d.est_band = 0;
    d.pack_bits = 0;
    d.bin_hz = (float)Fs / (float)Ndft;

    const int est_space = rc.est_space_rs * Rs;
    if (!fsk_est_range(Fs, Ndft, est_min, est_max, &d.est_st, &d.est_en)) return PIRIP_ERR_BAD_CONFIG;
    d.f_zero = (est_space * Ndft) / Fs;

    // Hann window by the recursive oscillator of fsk_generate_hann_table()
    hann.resize(Ndft);
    {
        const float w = rc.hann_denominator_ndft ? (2 * M_PI) / ((float)Ndft) : (2 * M_PI) / ((float)Ndft - 1);
        cf dphi{cosf(w), sinf(w)};
        cf rphi{.5f, 0.0f};
        rphi = cmul(cf{dphi.re, -dphi.im}, rphi);
        for (int i = 0; i < Ndft; i++) {
            rphi = cmul(dphi, rphi);
            hann[i] = .5f - rphi.re;
        }
    }

    // FFT twiddles exactly as kiss_fft_alloc(): float of double cos/sin
    twiddle.resize(2 * (size_t)Ndft);
    for (int i = 0; i < Ndft; i++) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / Ndft;
        twiddle[2 * i] = (float)std::cos(phase);
        twiddle[2 * i + 1] = (float)std::sin(phase);
    }

    // factorisation: 4s first, then a single 2 (Ndft is a power of two)
    int radices[kMaxStages], ms[kMaxStages], nst = 0;
    {
        int n = Ndft;
        while (n > 1) {
            int p = (n % 4 == 0) ? 4 : 2;
            n /= p;
            radices[nst] = p; ms[nst] = n; nst++;
        }
    }
    d.nstages = nst;
    // top-down level l has fstride = product of radices above it; execute bottom-up
    {
        int fs = 1;
        int fstr[kMaxStages];
        for (int l = 0; l < nst; l++) { fstr[l] = fs; fs *= radices[l]; }
        for (int l = 0; l < nst; l++) {
            int src = nst - 1 - l;
            stages[l] = FftStage{radices[src], ms[src], fstr[src]};
        }
        // leaf permutation: slot n = sum_l q_l*ms[l]  reads input index sum_l q_l*fstr[l]
        leaf_perm.resize(Ndft);
        for (int n = 0; n < Ndft; n++) {
            int rem = n, idx = 0;
            for (int l = 0; l < nst; l++) {
                int q = rem / ms[l]; rem -= q * ms[l];
                idx += q * fstr[l];
            }
            leaf_perm[n] = (uint16_t)idx;
        }
        leaf_iperm.resize(Ndft);
        for (int n = 0; n < Ndft; n++) leaf_iperm[leaf_perm[n]] = (uint16_t)n;
    }

    // u8 -> float conversion table of the configured front end
    u8_lut.resize(256);
    for (int x = 0; x < 256; x++) {
        if (in_format == PIRIP_IN_CU8_CSDR) u8_lut[x] = ((float)x) / (255 / 2.0) - 1.0;   // convert_u8_f
        else u8_lut[x] = ((float)x - (double)rc.u8d_offset) / (double)rc.u8d_scale;       // fsk_demod -d: (x - 127.0) / 128.0 as recalled (double, rounded once)
    }

    // fine-timing phasors exp(+j 2 pi k / P), double-rounded (the product's own choice: the
    // upstream recursion drifts; magnitudes agree to ~1e-6, see DESIGN.md tolerance table)
    timing_ph.resize(2 * (size_t)P);
    for (int k = 0; k < P; k++) {
        timing_ph[2 * k] = (float)std::cos(2.0 * M_PI * k / P);
        timing_ph[2 * k + 1] = (float)std::sin(2.0 * M_PI * k / P);
    }

    // mask estimator comb: 3-bin teeth at multiples of tone_spacing
    teeth.clear();
    mask_dtheta.assign(kMaxTones, 0u);
    d.mask_len = 0; d.n_teeth = 0;
    if (d.freq_est_type) {
        if (tone_spacing <= 0) return PIRIP_ERR_BAD_CONFIG;
        std::vector<uint8_t> mask(Ndft, 0);
        for (int i = 0; i < 3; i++) mask[i] = 1;
        int bin = 0;
        for (int m = 1; m <= M - 1; m++) {
            bin = (int)(std::round((double)((float)m * tone_spacing * Ndft / Fs)) - 1);
            for (int i = bin; i <= bin + 2; i++) if (i >= 0 && i < Ndft) mask[i] = 1;
        }
        d.mask_len = bin + 2 + 1;
        for (int i = 0; i < d.mask_len && i < Ndft; i++) if (mask[i]) teeth.push_back((int16_t)i);
        d.n_teeth = (int)teeth.size();
        mask_dtheta.assign((size_t)Ndft * M, 0u);                       // filled with the oscillator models below
    }
    if (teeth.empty()) teeth.push_back(0);

    // ---- upstream recursion models ---------------------------------------------------------
    {
        const int per = d.freq_est_type ? M : 1;
        osc_drift.assign(2 * (size_t)Ndft * per, 0.f);
        osc_step.assign(2 * (size_t)Ndft * per, 0.f);
        for (int b = 0; b < Ndft; b++) {
            for (int m = 0; m < per; m++) {
                float f_est; double ideal_turns;
                if (d.freq_est_type) {
                    // upstream's tone estimate is an INTEGER-truncated comb frequency plus m * spacing -- up to a bin fraction
                    // away from the comb position -- so the device phase accumulator is given the angle of the very
                    // (cosf, sinf) step upstream multiplies by (set below, once c and s are known)
                    const float foff = (float)((b - Ndft / 2) * Fs / Ndft);
                    f_est = foff + (float)(m * tone_spacing);
                    ideal_turns = 0.0;
                } else {
                    f_est = (float)(b - Ndft / 2) * d.bin_hz;
                    ideal_turns = (double)(b - Ndft / 2) / Ndft;
                }
                const float w = 2 * M_PI * ((f_est) / (float)(Fs));      // as fsk_demod_core computes dphi_m
                const float c = cosf(w), s = sinf(w);
                if (d.freq_est_type) {
                    double turns = std::atan2((double)s, (double)c) / (2.0 * M_PI);
                    turns -= std::floor(turns);
                    const uint32_t step = (uint32_t)((uint64_t)std::llround(turns * 4294967296.0) & 0xffffffffull);
                    mask_dtheta[(size_t)b * M + m] = step;
                    ideal_turns = (double)step / 4294967296.0;
                }
                const double a = std::sqrt((double)c * c + (double)s * s) - 1.0;
                double dd = std::atan2((double)s, (double)c) - 2.0 * M_PI * ideal_turns;
                dd -= 2.0 * M_PI * std::round(dd / (2.0 * M_PI));
                const size_t ix = 2 * ((size_t)b * per + m);
                osc_drift[ix] = (float)a; osc_drift[ix + 1] = (float)dd;
                osc_step[ix] = c; osc_step[ix + 1] = s;
            }
        }
        timing_rec.resize(2 * (size_t)d.nint);
        const float wt = 2 * M_PI * ((float)(Rs) / (float)(P * Rs));
        cf dph{cosf(wt), sinf(wt)}, ph{1.0f, 0.0f};
        for (int i = 0; i < d.nint; i++) {
            timing_rec[2 * i] = ph.re; timing_rec[2 * i + 1] = ph.im;
            ph = cmul(ph, dph);
        }
    }
    // ---- fast-kernel tables (16 points per lane, Ndft == 256) -------------------------------
    fast_tab.clear();
    for (auto &v : tw_s2) v = 0.f;
    if (Ndft == 256) {
        fast_tab.assign(16 * 48, 0.f);
        for (int e = 0; e < 16; e++) {
            float *row = &fast_tab[(size_t)e * 48];
            const int a = e >> 2, b = e & 3, base = a + 4 * b;      // stage-1/2 lane g = 4a+b
            for (int t = 0; t < 16; t++) row[t] = hann[base + 16 * t];
            for (int r = 1; r <= 3; r++) {                          // stage 3: k = e, fstride 4
                row[16 + 2 * (r - 1)] = twiddle[2 * (4 * e * r)];
                row[16 + 2 * (r - 1) + 1] = twiddle[2 * (4 * e * r) + 1];
            }
            for (int bp = 0; bp < 4; bp++)                          // stage 4: k = e + 16 b', fstride 1
                for (int r = 1; r <= 3; r++) {
                    const int k = (e + 16 * bp) * r;
                    row[22 + 2 * (3 * bp + (r - 1))] = twiddle[2 * k];
                    row[22 + 2 * (3 * bp + (r - 1)) + 1] = twiddle[2 * k + 1];
                }
        }
        for (int k = 1; k <= 3; k++)
            for (int r = 1; r <= 3; r++) {
                tw_s2[2 * (3 * (k - 1) + (r - 1))] = twiddle[2 * (16 * k * r)];
                tw_s2[2 * (3 * (k - 1) + (r - 1)) + 1] = twiddle[2 * (16 * k * r) + 1];
            }
    }
    // ---- wave kernel tables for Ndft == 512 (kiss_fft factors 4,4,4,4,2; see fsk_demod_wave.hip) ---------
    //   [0, 512)          Hann window
    //   [512, 768)        per r2 = 0..7, 16 complex: tw[16 r2 j] (j = 1..3), then tw[4 (r2 + 8 j2) j] for j2 = 0..3, j = 1..3
    //   [768, 1536)       [3 jj + (j-1)][L] complex, L = 0..31: tw[(L + 32 jj) j]
    //   tw_s2[0..5]       tw[64], tw[128], tw[192] (the radix-4 m = 2 level's only non-trivial twiddles)
    if (Ndft == 512) {
        fast_tab.assign(512 + 8 * 16 * 2 + 12 * 32 * 2, 0.f);
        for (int i = 0; i < 512; i++) fast_tab[i] = hann[i];
        float *p2 = &fast_tab[512];
        for (int r2 = 0; r2 < 8; r2++) {
            for (int j = 1; j <= 3; j++) {
                p2[2 * (r2 * 16 + (j - 1))] = twiddle[2 * (16 * r2 * j)];
                p2[2 * (r2 * 16 + (j - 1)) + 1] = twiddle[2 * (16 * r2 * j) + 1];
            }
            for (int j2 = 0; j2 < 4; j2++)
                for (int j = 1; j <= 3; j++) {
                    const int k = 4 * (r2 + 8 * j2) * j;
                    p2[2 * (r2 * 16 + 3 + 3 * j2 + (j - 1))] = twiddle[2 * k];
                    p2[2 * (r2 * 16 + 3 + 3 * j2 + (j - 1)) + 1] = twiddle[2 * k + 1];
                }
        }
        float *p3 = &fast_tab[512 + 8 * 16 * 2];
        for (int jj = 0; jj < 4; jj++)
            for (int j = 1; j <= 3; j++)
                for (int L = 0; L < 32; L++) {
                    const int k = (L + 32 * jj) * j;
                    p3[2 * ((3 * jj + (j - 1)) * 32 + L)] = twiddle[2 * k];
                    p3[2 * ((3 * jj + (j - 1)) * 32 + L) + 1] = twiddle[2 * k + 1];
                }
        for (int j = 1; j <= 3; j++) { tw_s2[2 * (j - 1)] = twiddle[2 * (64 * j)]; tw_s2[2 * (j - 1) + 1] = twiddle[2 * (64 * j) + 1]; }
    }
    // ---- wave kernel tables for Ndft == 128 (kiss_fft factors 4,4,4,2: radix-2 m=1, radix-4 m=2, 8, 32; see fsk_demod_wave.hip) --
    //   [0, 128)          Hann window
    //   [128, 384)        per r = 0..7, 16 complex: tw[4 r j] (j = 1..3: level m = 8), then tw[(r + 8 j1) j] for j1 = 0..3, j = 1..3 (m = 32)
    //   tw_s2[0..5]       tw[16], tw[32], tw[48] (the radix-4 m = 2 level's only non-trivial twiddles)
    if (Ndft == 128) {
        fast_tab.assign(128 + 8 * 16 * 2, 0.f);
        for (int i = 0; i < 128; i++) fast_tab[i] = hann[i];
        float *p2 = &fast_tab[128];
        for (int r = 0; r < 8; r++) {
            for (int j = 1; j <= 3; j++) {
                p2[2 * (r * 16 + (j - 1))] = twiddle[2 * (4 * r * j)];
                p2[2 * (r * 16 + (j - 1)) + 1] = twiddle[2 * (4 * r * j) + 1];
            }
            for (int j1 = 0; j1 < 4; j1++)
                for (int j = 1; j <= 3; j++) {
                    const int k = (r + 8 * j1) * j;
                    p2[2 * (r * 16 + 3 + 3 * j1 + (j - 1))] = twiddle[2 * k];
                    p2[2 * (r * 16 + 3 + 3 * j1 + (j - 1)) + 1] = twiddle[2 * k + 1];
                }
        }
        for (int j = 1; j <= 3; j++) { tw_s2[2 * (j - 1)] = twiddle[2 * (16 * j)]; tw_s2[2 * (j - 1) + 1] = twiddle[2 * (16 * j) + 1]; }
    }
    // Ndft == 4096 (fsk_demod_block.hip): the last register pass's twiddles as [16 rows][256 threads] complex, row i < 3: tw[4 t (i + 1)]
    // (stage m = 256), row 3 + 3 k4 + (r - 1): tw[(t + 256 k4) r] (stage m = 1024), row 15 unused -- one wave-uniform row base plus the
    // thread's own offset addresses every load
    if (Ndft == 4096) {
        fast_tab.assign(16 * 256 * 2, 0.f);
        for (int i = 0; i < 15; i++)
            for (int t = 0; t < 256; t++) {
                const int k4 = i < 3 ? 0 : (i - 3) / 3, r = i < 3 ? i + 1 : (i - 3) % 3 + 1;
                const int k = i < 3 ? 4 * t * r : (t + 256 * k4) * r;
                fast_tab[2 * (i * 256 + t)] = twiddle[2 * k]; fast_tab[2 * (i * 256 + t) + 1] = twiddle[2 * k + 1];
            }
    }
    return PIRIP_OK;
}

bool fsk_est_range(int Fs, int Ndft, int est_min, int est_max, int *st_out, int *en_out)
{
    if (est_min == 0 && est_max == 0) { est_min = 0; est_max = Fs; }   // fsk_create defaults
    else if (est_min < -Fs / 2 || est_max > Fs / 2 || est_max <= est_min) return false;   // fsk_set_freq_est_limits() asserts
    int st = (est_min * Ndft) / Fs + Ndft / 2; if (st < 0) st = 0;
    int en = (est_max * Ndft) / Fs + Ndft / 2; if (en > Ndft) en = Ndft;
    *st_out = st; *en_out = en;
    return true;
}

// ------------------------------------------------------------------------------------------
// Tx side: continuous-phase M-FSK [UPSTREAM-RECALLED codec2 fsk.c: fsk_mod / fsk_mod_c].
// Bits MSB-first per symbol, higher symbol = higher tone: /root/reference/tx/rpitx_fsk.cpp:129-141
// ------------------------------------------------------------------------------------------
void FskMod::init(int Fs_, int Rs_, int M_, int f1, int spacing)
{
    Fs = Fs_; Rs = Rs_; M = M_; Ts = Fs_ / Rs_; f1_tx = f1; tone_spacing = spacing;
    ph_re = cosf(0); ph_im = sinf(0);
}

void FskMod::mod(const uint8_t *bits, int nbits, float *out, bool complex_out)
{
    cf dosc[kMaxTones];
    for (int m = 0; m < M; m++) {
        const float w = 2 * M_PI * ((float)(f1_tx + (tone_spacing * m)) / (float)(Fs));
        dosc[m] = cf{cosf(w), sinf(w)};
    }
    const int bps = (M == 2) ? 1 : 2;
    const int nsym = nbits / bps;
    cf ph{ph_re, ph_im};
    int bit_i = 0;
    for (int i = 0; i < nsym; i++) {
        int sym = 0;
        for (int b = 0; b < bps; b++) sym = (sym << 1) | (bits[bit_i++] == 1 ? 1 : 0);
        const cf dph = dosc[sym];
        for (int j = 0; j < Ts; j++) {
            ph = cmul(ph, dph);
            if (complex_out) { out[2 * (i * Ts + j)] = 2 * ph.re; out[2 * (i * Ts + j) + 1] = 2 * ph.im; }
            else out[i * Ts + j] = 2 * ph.re;
        }
    }
    const float av = sqrtf((ph.re * ph.re) + (ph.im * ph.im));
    ph_re = ph.re / av; ph_im = ph.im / av;
}

// 100-bit pseudo-random test frame [UPSTREAM-RECALLED codec2 fsk_get_test_bits.c]; packet size
// pinned by /root/reference/test/include.sh:7. glibc rand(), seed unverified.
void test_frame_bits(uint8_t *frame, int framesize)
{
    srand(158324);
    for (int i = 0; i < framesize; i++) frame[i] = rand() & 0x1;
}

void PutBits::init(int framesize_, float valid_thresh_)
{
    framesize = framesize_; valid_thresh = valid_thresh_;
    tx.resize(framesize); rx.assign(framesize, 0);
    test_frame_bits(tx.data(), framesize);
    bitcnt = biterr = 0; packetcnt = 0;
}

bool PutBits::push(uint8_t bit, int *errs_out)
{
    rx[framesize - 1] = bit;
    int errs = 0;
    for (int i = 0; i < framesize; i++) errs += rx[i] != tx[i];
    bool valid = errs < valid_thresh * framesize;
    if (valid) { packetcnt++; bitcnt += framesize; biterr += errs; }
    std::memmove(rx.data(), rx.data() + 1, framesize - 1);
    if (errs_out) *errs_out = errs;
    return valid;
}

// csdr low-pass design ---------------------------------------------------------------------
#define CSDR_PI ((float)3.14159265358979323846)
int csdr_filter_len(float transition_bw)
{
    int result = 4.0 / transition_bw;
    if (result % 2 == 0) result++;
    return result;
}
static float hamming_kernel(float rate)
{
    rate = 0.5 + rate / 2;
    return 0.54 - 0.46 * std::cos((double)(2 * CSDR_PI * rate));   // C: cos() on a float argument is the double cos
}
static float blackman_kernel(float rate)
{
    rate = 0.5 + rate / 2;
    return 0.42 - 0.5 * std::cos((double)(2 * CSDR_PI * rate)) + 0.08 * std::cos((double)(4 * CSDR_PI * rate));
}
static float window_kernel(int window, float rate)
{
    return window == 0 ? 1.0f : window == 1 ? blackman_kernel(rate) : hamming_kernel(rate);
}
void csdr_lowpass_hamming(float *taps, int length, float cutoff_rate) { csdr_lowpass(taps, length, cutoff_rate, 2); }
void csdr_lowpass(float *taps, int length, float cutoff_rate, int window)
{
    const int middle = length / 2;
    taps[middle] = 2 * CSDR_PI * cutoff_rate * window_kernel(window, 0);
    for (int i = 1; i <= middle; i++)
        taps[middle - i] = taps[middle + i] =
            (std::sin((double)(2 * CSDR_PI * cutoff_rate * i)) / i) * window_kernel(window, (float)i / middle);
    float sum = 0;
    for (int i = 0; i < length; i++) sum += taps[i];
    for (int i = 0; i < length; i++) taps[i] /= sum;
}

// PIRIP_RECALLED="field=value,field=value" over what r holds (the drill's switch: the command-line tools and pirip_hip_create read it,
// the checker's pin_against_ref.py says which field to try: INTEGRATION.md 4); false on a name that is not a field
bool recalled_from_env(pirip_fsk_recalled *r)
{
    const char *e = getenv("PIRIP_RECALLED");
    if (!e) return true;
    std::string all(e);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string tok = all.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = tok.find('=');
        if (eq == std::string::npos) return false;
        const std::string k = tok.substr(0, eq);
        const double v = atof(tok.c_str() + eq + 1);
        if (k == "hann_denominator_ndft") r->hann_denominator_ndft = (int)v;
        else if (k == "tc") r->tc = (float)v;
        else if (k == "est_space_rs") r->est_space_rs = (float)v;
        else if (k == "nin_threshold") r->nin_threshold = (float)v;
        else if (k == "nin_step_div") r->nin_step_div = (int)v;
        else if (k == "s16_scale") r->s16_scale = (float)v;
        else if (k == "u8d_offset") r->u8d_offset = (float)v;
        else if (k == "u8d_scale") r->u8d_scale = (float)v;
        else if (k == "ndft_rule") r->ndft_rule = (int)v;
        else if (k == "sf_power") r->sf_power = (int)v;
        else return false;
    }
    return true;
}

void recalled_defaults(pirip_fsk_recalled *r)
{
    r->hann_denominator_ndft = 0;
    r->tc = 0.1f;
    r->est_space_rs = 0.75f;
    r->nin_threshold = 0.25f;
    r->nin_step_div = 4;
    r->s16_scale = (float)PIRIP_FDMDV_SCALE;
    r->u8d_offset = 127.0f;
    r->u8d_scale = 128.0f;
    r->ndft_rule = 0;
    r->sf_power = 0;
}

}  // namespace pirip
