// pirip_amd/csrc/codec2_shim.hip -- include/pirip_hip.h section C: codec2's single-stream FSK
// API [UPSTREAM-RECALLED codec2 src/fsk.h] served by the HIP demodulator, so that programs
// written against libcodec2 (fsk_demod.c, rtl_fsk.c -- linked by /root/reference/build_rtlsdr.sh:9)
// relink against libpirip_hip.so unchanged. Conventions kept: opaque handle, caller owns
// every buffer, no error codes (codec2 asserts -> we print and abort), the caller re-queries
// fsk_nin() before every fsk_demod(). One handle = one device-resident stream; every
// fsk_demod() is an upload + one-frame launch + download, i.e. the correctness boundary, not
// the throughput path (that is section A with many streams).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pirip_hip.h"
#include "fsk_plan.hpp"

using namespace pirip;

struct FSK {
    int burst = 0;
    pirip_fsk_params prm;
    pirip_hip_demod *dev = nullptr;
    pirip_fsk_info info{};
    FskMod mod;
    int f1_tx = 0, tone_spacing = 0;
    int nin = 0, Ndft = 0;
    float last[8] = {0};
    std::vector<uint8_t> bits;
    std::vector<float> filt;
};

namespace {

[[noreturn]] void die(const char *what, int rc)
{
    fprintf(stderr, "libpirip_hip (codec2 shim): %s: %s -- this build has no CPU fallback\n", what, pirip_hip_strerror(rc));
    abort();
}

void ensure_device(struct FSK *f)
{
    if (f->dev) return;
    int rc = pirip_hip_create(&f->prm, 1, -1, &f->dev);
    if (rc != PIRIP_OK) die("pirip_hip_create", rc);
    pirip_hip_get_info(f->dev, &f->info);
    f->nin = f->info.N;
    if (f->burst) pirip_hip_set_burst_mode(f->dev, 1);
}

void run(struct FSK *f, uint8_t *rx_bits, float *rx_filt, COMP *in)
{
    ensure_device(f);
    int64_t nf = 0, cons = 0;
    float st[PIRIP_STATS_PER_FRAME];
    int rc = pirip_hip_demod_host(f->dev, in, f->nin, f->bits.data(), f->filt.data(), st, 1, &nf, &cons);
    if (rc != PIRIP_OK) die("pirip_hip_demod_host", rc);
    if (nf == 1) {
        if (rx_bits) memcpy(rx_bits, f->bits.data(), f->info.Nbits);
        if (rx_filt) memcpy(rx_filt, f->filt.data(), sizeof(float) * f->prm.M * f->prm.Nsym);
        memcpy(f->last, st, sizeof(st));
    }
    f->nin = pirip_hip_nin0(f->dev);
}

}  // namespace

extern "C" {

struct FSK *fsk_create_hbr(int Fs, int Rs, int M, int P, int Nsym, int f1_tx, int tone_spacing)
{
    FskPlan probe;
    // codec2 asserts on these; report and abort the same way
    int rc = probe.init(Fs, Rs, M, P, Nsym, 0, 0, 0, tone_spacing, PIRIP_IN_CF32);
    if (rc != PIRIP_OK) die("fsk_create_hbr", rc);
    struct FSK *f = new FSK();
    f->prm = pirip_fsk_params{Fs, Rs, M, P, Nsym, 0, 0, 0, tone_spacing, PIRIP_IN_CF32};
    f->f1_tx = f1_tx; f->tone_spacing = tone_spacing;
    f->nin = probe.d.N; f->Ndft = probe.d.Ndft;
    f->info.Ts = probe.d.Ts; f->info.N = probe.d.N; f->info.Nmem = probe.d.Nmem; f->info.Ndft = probe.d.Ndft;
    f->info.Nbits = probe.d.Nbits; f->info.nin_max = probe.d.N + probe.d.Ts / 4;
    f->bits.resize(probe.d.Nbits); f->filt.resize((size_t)M * Nsym);
    f->mod.init(Fs, Rs, M, f1_tx, tone_spacing);
    return f;
}

struct FSK *fsk_create(int Fs, int Rs, int M, int tx_f1, int tx_fs)
{
    return fsk_create_hbr(Fs, Rs, M, PIRIP_FSK_DEFAULT_P, PIRIP_FSK_DEFAULT_NSYM, tx_f1, tx_fs);
}

void fsk_destroy(struct FSK *f)
{
    if (!f) return;
    if (f->dev) pirip_hip_destroy(f->dev);
    delete f;
}

void fsk_set_freq_est_limits(struct FSK *f, int est_min, int est_max)
{
    f->prm.est_min = est_min; f->prm.est_max = est_max;
    FskPlan probe;
    int rc = probe.init(f->prm.Fs, f->prm.Rs, f->prm.M, f->prm.P, f->prm.Nsym, est_min, est_max,
                        f->prm.freq_est_type, f->prm.tone_spacing, PIRIP_IN_CF32);
    if (rc != PIRIP_OK) die("fsk_set_freq_est_limits", rc);
    if (f->dev) { pirip_hip_destroy(f->dev); f->dev = nullptr; }   // re-planned on next demod
}

void fsk_set_freq_est_alg(struct FSK *f, int est_type)
{
    f->prm.freq_est_type = est_type ? 1 : 0;
    if (f->dev) { pirip_hip_destroy(f->dev); f->dev = nullptr; }
}

uint32_t fsk_nin(struct FSK *f) { return (uint32_t)f->nin; }
void fsk_demod(struct FSK *f, uint8_t rx_bits[], COMP fsk_in[]) { run(f, rx_bits, nullptr, fsk_in); }
void fsk_demod_sd(struct FSK *f, float rx_filt[], COMP fsk_in[]) { run(f, nullptr, rx_filt, fsk_in); }

void fsk_clear_estimators(struct FSK *f)
{
    // upstream zeroes Sf and resets nin; a device reset also clears the oscillator phases and
    // the integrator memory, which only matters for the first symbols after the call
    if (f->dev) { int rc = pirip_hip_reset(f->dev, nullptr); if (rc != PIRIP_OK) die("pirip_hip_reset", rc); }
    f->nin = f->info.N;
}

void fsk_enable_burst_mode(struct FSK *f)
{
    f->burst = 1;
    f->nin = f->info.N;
    if (f->dev) { int rc = pirip_hip_set_burst_mode(f->dev, 1); if (rc != PIRIP_OK) die("pirip_hip_set_burst_mode", rc); }
}

void fsk_get_demod_stats(struct FSK *f, struct MODEM_STATS *st)
{
    memset(st, 0, sizeof(*st));
    st->Nc = f->prm.M;
    st->rx_timing = f->last[4] * (float)f->prm.P;
    st->clock_offset = f->last[7];
    for (int m = 0; m < f->prm.M; m++) st->f_est[m] = f->last[m];
    // snr_est (smoothed EbNodB) and foff live in the device-side scalars
    if (f->dev) {
        float s8[8];
        if (pirip_hip_get_scalars(f->dev, 0, s8) == PIRIP_OK) st->clock_offset = s8[7];
        st->snr_est = f->last[5] > 0.f ? 10.0f * log10f(f->last[5]) : 0.f;
    }
    const float fc_avg = (st->f_est[0] + st->f_est[f->prm.M - 1]) / 2;
    const float fc_tx = (float)(f->f1_tx + f->f1_tx + f->tone_spacing * (f->prm.M - 1)) / 2;
    st->foff = fc_tx - fc_avg;
}

void fsk_stats_normalise_eye(struct FSK *, int) {}

void fsk_mod(struct FSK *f, float fsk_out[], uint8_t tx_bits[], int nbits) { f->mod.mod(tx_bits, nbits, fsk_out, false); }
void fsk_mod_c(struct FSK *f, COMP fsk_out[], uint8_t tx_bits[], int nbits) { f->mod.mod(tx_bits, nbits, (float *)fsk_out, true); }

int fsk_get_Nbits(struct FSK *f) { return f->info.Nbits; }
int fsk_get_Nsym(struct FSK *f) { return f->prm.Nsym; }
int fsk_get_N(struct FSK *f) { return f->info.N; }
int fsk_get_Ts(struct FSK *f) { return f->info.Ts; }
int fsk_get_Ndft(struct FSK *f) { return f->Ndft; }
float fsk_get_norm_rx_timing(struct FSK *f) { return f->last[4]; }
float fsk_get_SNRest(struct FSK *f) { return f->last[5]; }
void fsk_get_f_est(struct FSK *f, float f_est[]) { for (int m = 0; m < f->prm.M; m++) f_est[m] = f->last[m]; }
void fsk_get_Sf(struct FSK *f, float Sf[])
{
    ensure_device(f);
    int rc = pirip_hip_get_Sf(f->dev, 0, Sf);
    if (rc != PIRIP_OK) die("pirip_hip_get_Sf", rc);
}

}  // extern "C"
