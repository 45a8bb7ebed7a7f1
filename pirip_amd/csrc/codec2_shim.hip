// pirip_amd/csrc/codec2_shim.hip -- include/pirip_hip.h section C: codec2's single-stream FSK API
// [UPSTREAM-RECALLED codec2 src/fsk.h, fsk.c, modem_stats.h] served by the HIP demodulator, so that programs written
// against libcodec2 (fsk_demod.c, rtl_fsk.c -- linked by /root/reference/build_rtlsdr.sh:9) rebuild against
// include/pirip_hip.h + libpirip_hip.so. Conventions kept: handle created/destroyed by the library, caller owns every
// sample/bit buffer, no error codes (codec2 asserts -> we print and abort), the caller re-queries fsk_nin() before every
// fsk_demod(), and the fields those programs read straight out of struct FSK (Nbits, Ndft, nin, f_est[], f2_est[],
// norm_rx_timing, SNRest, EbNodB, ppm, v_est, Sf[]) are public and refreshed after every demodulator call.
// One handle = one device-resident stream; every fsk_demod() is an upload + one-frame launch + download, i.e. the
// correctness boundary, not the throughput path (that is section A with many streams).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <dlfcn.h>
#include <string>
#include <sys/stat.h>

#include "../../include/pirip_hip.h"
#include "fsk_ldpc.hpp"
#include "fsk_plan.hpp"

using namespace pirip;

namespace {

struct Priv {
    pirip_fsk_params prm{};
    pirip_hip_demod *dev = nullptr;
    pirip_fsk_info info{};
    FskMod mod;
    bool ran = false;                  // a frame has been demodulated: estimator state exists on the device
    bool want_eye = false;             // MODEM_STATS.rx_eye asked for (fsk_stats_normalise_eye / PIRIP_SHIM_EYE): any-configuration kernel
    std::vector<uint8_t> bits;
    std::vector<float> filt, Sf;
};

Priv *P(struct FSK *f) { return (Priv *)f->pirip_priv; }

[[noreturn]] void die(const char *what, int rc)
{
    fprintf(stderr, "libpirip_hip (codec2 shim): %s: %s -- this build has no CPU fallback\n", what, pirip_hip_strerror(rc));
    abort();
}

void ensure_device(struct FSK *f)
{
    Priv *p = P(f);
    if (p->dev) return;
    int rc = pirip_hip_create(&p->prm, 1, -1, &p->dev);
    if (rc != PIRIP_OK) die("pirip_hip_create", rc);
    pirip_hip_get_info(p->dev, &p->info);
    f->nin = p->info.N;
    if (f->burst_mode) pirip_hip_set_burst_mode(p->dev, 1);
    // codec2 (not built __EMBEDDED__) keeps the eye diagram of every frame in fsk->stats. Here that is a diagnostic only the
    // any-configuration kernel can write (pirip_hip_enable_eye: "not for throughput"), so a handle stays on its specialised
    // wave / block instance until the program shows that it wants the eye: fsk_stats_normalise_eye(), or PIRIP_SHIM_EYE=1.
    if (p->want_eye) {
        rc = pirip_hip_enable_eye(p->dev, 1);
        if (rc != PIRIP_OK) die("pirip_hip_enable_eye", rc);
    }
}

void ask_for_eye(struct FSK *f)
{
    Priv *p = P(f);
    if (p->want_eye) return;
    p->want_eye = true;
    if (!p->dev) return;               // taken up at ensure_device(): nothing has run, nothing is lost
    if (p->ran)
        fprintf(stderr, "libpirip_hip (codec2 shim): eye diagram asked for mid-stream: the handle moves to the any-configuration kernel and "
                        "its demodulator state restarts (call fsk_stats_normalise_eye() before the first fsk_demod(), or set "
                        "PIRIP_SHIM_EYE=1, to avoid this)\n");
    int rc = pirip_hip_enable_eye(p->dev, 1);
    if (rc != PIRIP_OK) die("pirip_hip_enable_eye", rc);
    p->ran = false;
    f->nin = p->info.N;
}

// mirror the device-side stream state into the public fields
void refresh(struct FSK *f, const float *st /* per-frame stats of the frame just run, or NULL */)
{
    Priv *p = P(f);
    pirip_stream_state s;
    int rc = pirip_hip_get_stream_state(p->dev, 0, &s);
    if (rc != PIRIP_OK) die("pirip_hip_get_stream_state", rc);
    f->nin = s.nin; f->norm_rx_timing = s.norm_rx_timing; f->ppm = s.ppm; f->SNRest = s.SNRest;
    f->EbNodB = s.EbNodB; f->v_est = s.v_est; f->rx_sig_pow = s.rx_sig_pow; f->rx_nse_pow = s.rx_nse_pow;
    for (int m = 0; m < MODE_M_MAX; m++) { f->f_est[m] = s.f_est[m]; f->f2_est[m] = s.f_est[m]; }
    if (f->stats) f->stats->snr_est = s.snr_est;
    (void)st;
    rc = pirip_hip_get_Sf(p->dev, 0, p->Sf.data());
    if (rc != PIRIP_OK) die("pirip_hip_get_Sf", rc);
    if (f->stats && p->ran && p->want_eye) {
        rc = pirip_hip_get_eye(p->dev, 0, f->normalise_eye, &f->stats->rx_eye[0][0], &f->stats->neyetr, &f->stats->neyesamp);
        if (rc != PIRIP_OK) die("pirip_hip_get_eye", rc);
    }
}

void run(struct FSK *f, uint8_t *rx_bits, float *rx_filt, COMP *in)
{
    ensure_device(f);
    Priv *p = P(f);
    int64_t nf = 0, cons = 0;
    float st[PIRIP_STATS_PER_FRAME];
    int rc = pirip_hip_demod_host(p->dev, in, f->nin, p->bits.data(), p->filt.data(), st, 1, &nf, &cons);
    if (rc != PIRIP_OK) die("pirip_hip_demod_host", rc);
    if (nf == 1) {
        if (rx_bits) memcpy(rx_bits, p->bits.data(), (size_t)p->info.Nbits);
        if (rx_filt) memcpy(rx_filt, p->filt.data(), sizeof(float) * (size_t)p->prm.M * p->prm.Nsym);
        p->ran = true;
    }
    refresh(f, st);
}

// a new plan for changed estimator settings; state that survives (Sf, timing, ppm) is only lost when no frame has run yet
void replan(struct FSK *f)
{
    Priv *p = P(f);
    if (p->dev) { pirip_hip_destroy(p->dev); p->dev = nullptr; }
    if (p->ran) {
        fprintf(stderr, "libpirip_hip (codec2 shim): estimator algorithm changed mid-stream: demodulator state restarts "
                        "(codec2 keeps Sf; call fsk_set_freq_est_alg before the first fsk_demod to avoid this)\n");
        p->ran = false;
    }
}

}  // namespace

extern "C" {

struct FSK *fsk_create_hbr(int Fs, int Rs, int M, int P_, int Nsym, int f1_tx, int tone_spacing)
{
    FskPlan probe;
    // codec2 asserts on these; report and abort the same way
    int rc = probe.init(Fs, Rs, M, P_, Nsym, 0, 0, 0, tone_spacing, PIRIP_IN_CF32);
    if (rc != PIRIP_OK) die("fsk_create_hbr", rc);
    struct FSK *f = (struct FSK *)calloc(1, sizeof(struct FSK));
    Priv *p = new Priv();
    f->pirip_priv = p;
    p->prm = pirip_fsk_params{Fs, Rs, M, P_, Nsym, 0, 0, 0, tone_spacing, PIRIP_IN_CF32};
    const FskDims &d = probe.d;
    f->Ndft = d.Ndft; f->Fs = Fs; f->N = d.N; f->Rs = Rs; f->Ts = d.Ts; f->Nmem = d.Nmem; f->P = P_; f->Nsym = Nsym; f->Nbits = d.Nbits;
    f->f1_tx = f1_tx; f->tone_spacing = tone_spacing; f->mode = M; f->tc = d.tc;
    f->est_min = 0; f->est_max = Fs; f->est_space = (int)(0.75 * Rs);
    f->nin = d.N; f->tx_phase_c.real = 1.0f;
    p->info.Ts = d.Ts; p->info.N = d.N; p->info.Nmem = d.Nmem; p->info.Ndft = d.Ndft; p->info.Nbits = d.Nbits; p->info.nin_max = d.N + d.nin_step;
    p->bits.resize((size_t)d.Nbits); p->filt.resize((size_t)M * Nsym); p->Sf.assign((size_t)d.Ndft, 0.f);
    f->Sf = p->Sf.data();
    f->stats = (struct MODEM_STATS *)calloc(1, sizeof(struct MODEM_STATS));
    f->normalise_eye = 1;              // [UPSTREAM-RECALLED fsk.c fsk_create_core]
    const char *ev = getenv("PIRIP_SHIM_EYE");
    p->want_eye = ev && *ev && *ev != '0';
    p->mod.init(Fs, Rs, M, f1_tx, tone_spacing);
    return f;
}

struct FSK *fsk_create(int Fs, int Rs, int M, int tx_f1, int tx_fs)
{
    return fsk_create_hbr(Fs, Rs, M, PIRIP_FSK_DEFAULT_P, PIRIP_FSK_DEFAULT_NSYM, tx_f1, tx_fs);
}

void fsk_destroy(struct FSK *f)
{
    if (!f) return;
    Priv *p = P(f);
    if (p->dev) pirip_hip_destroy(p->dev);
    delete p;
    free(f->stats);
    free(f);
}

void fsk_set_freq_est_limits(struct FSK *f, int est_min, int est_max)
{
    Priv *p = P(f);
    int st, en;
    if (!fsk_est_range(f->Fs, f->Ndft, est_min, est_max, &st, &en)) die("fsk_set_freq_est_limits", PIRIP_ERR_BAD_CONFIG);
    p->prm.est_min = est_min; p->prm.est_max = est_max;
    f->est_min = est_min; f->est_max = est_max;
    if (p->dev) { int rc = pirip_hip_set_freq_est_limits(p->dev, est_min, est_max); if (rc != PIRIP_OK) die("fsk_set_freq_est_limits", rc); }
}

void fsk_set_freq_est_alg(struct FSK *f, int est_type)
{
    Priv *p = P(f);
    const int t = est_type ? 1 : 0;
    if (t == p->prm.freq_est_type) return;
    p->prm.freq_est_type = t; f->freq_est_type = t;
    replan(f);
}

uint32_t fsk_nin(struct FSK *f) { return (uint32_t)f->nin; }
void fsk_demod(struct FSK *f, uint8_t rx_bits[], COMP fsk_in[]) { run(f, rx_bits, nullptr, fsk_in); }
void fsk_demod_sd(struct FSK *f, float rx_filt[], COMP fsk_in[]) { run(f, nullptr, rx_filt, fsk_in); }

void fsk_clear_estimators(struct FSK *f)
{
    // upstream zeroes Sf and resets nin; everything else (oscillator phases, integrator memory, timing) stays
    Priv *p = P(f);
    if (p->dev) { int rc = pirip_hip_clear_estimators(p->dev, nullptr); if (rc != PIRIP_OK) die("pirip_hip_clear_estimators", rc); }
    std::fill(p->Sf.begin(), p->Sf.end(), 0.f);
    f->nin = f->N;
}

void fsk_enable_burst_mode(struct FSK *f)
{
    Priv *p = P(f);
    f->burst_mode = 1;
    f->nin = f->N;
    if (p->dev) { int rc = pirip_hip_set_burst_mode(p->dev, 1); if (rc != PIRIP_OK) die("pirip_hip_set_burst_mode", rc); }
}

void fsk_get_demod_stats(struct FSK *f, struct MODEM_STATS *st)
{
    // [UPSTREAM-RECALLED fsk.c: fsk_get_demod_stats] snr_est is the smoothed EbNodB the demodulator maintains,
    // rx_timing / clock_offset / f_est copy the struct fields, foff = centre of the Tx tone plan - centre of the estimates
    const float snr = f->stats ? f->stats->snr_est : 0.f;
    if (st != f->stats) memset(st, 0, sizeof(*st));
    st->Nc = f->mode;
    st->snr_est = snr;
    st->rx_timing = f->norm_rx_timing * (float)f->P;
    st->clock_offset = f->ppm;
    for (int m = 0; m < f->mode; m++) st->f_est[m] = f->f_est[m];
    const float fc_avg = (st->f_est[0] + st->f_est[f->mode - 1]) / 2;
    const float fc_tx = (float)(f->f1_tx + f->f1_tx + f->tone_spacing * (f->mode - 1)) / 2;
    st->foff = fc_tx - fc_avg;
    // eye diagram of the latest frame (fsk->stats holds it, as upstream's does)
    if (f->stats && st != f->stats) {
        st->neyetr = f->stats->neyetr; st->neyesamp = f->stats->neyesamp;
        memcpy(st->rx_eye, f->stats->rx_eye, sizeof(st->rx_eye));
    }
}

// a program that sets how the eye is scaled is a program that reads it: this call is the shim's opt-in for the traces
void fsk_stats_normalise_eye(struct FSK *f, int enable) { f->normalise_eye = enable; ask_for_eye(f); }

void fsk_mod(struct FSK *f, float fsk_out[], uint8_t tx_bits[], int nbits)
{
    Priv *p = P(f);
    p->mod.mod(tx_bits, nbits, fsk_out, false);
    f->tx_phase_c.real = p->mod.ph_re; f->tx_phase_c.imag = p->mod.ph_im;
}
void fsk_mod_c(struct FSK *f, COMP fsk_out[], uint8_t tx_bits[], int nbits)
{
    Priv *p = P(f);
    p->mod.mod(tx_bits, nbits, (float *)fsk_out, true);
    f->tx_phase_c.real = p->mod.ph_re; f->tx_phase_c.imag = p->mod.ph_im;
}

int fsk_get_Nbits(struct FSK *f) { return f->Nbits; }
int fsk_get_Nsym(struct FSK *f) { return f->Nsym; }
int fsk_get_N(struct FSK *f) { return f->N; }
int fsk_get_Ts(struct FSK *f) { return f->Ts; }
int fsk_get_Ndft(struct FSK *f) { return f->Ndft; }
float fsk_get_norm_rx_timing(struct FSK *f) { return f->norm_rx_timing; }
float fsk_get_SNRest(struct FSK *f) { return f->SNRest; }
void fsk_get_f_est(struct FSK *f, float f_est[]) { for (int m = 0; m < f->mode; m++) f_est[m] = f->f_est[m]; }
void fsk_get_Sf(struct FSK *f, float Sf[])
{
    ensure_device(f);
    int rc = pirip_hip_get_Sf(P(f)->dev, 0, Sf);
    if (rc != PIRIP_OK) die("pirip_hip_get_Sf", rc);
}

}  // extern "C"

// ---- include/pirip_hip.h section F: the FreeDV API of FREEDV_MODE_FSK_LDPC [UPSTREAM-RECALLED codec2 freedv_api.c / freedv_fsk.c] --------
// What upstream's rtl_fsk.c binds in --code mode and what /root/reference/tx/rpitx_fsk.cpp:164-165,222,319-325,541 binds on the Tx side.
// One struct freedv = one struct FSK of section C (its device demodulator) + one section E receiver; freedv_rawdatacomprx is
// pirip_hip_fsk_ldpc_rx_batch on one staged frame (upload, one call's worth of kernels, 1 + k/8 + info bytes back): the drop-in
// boundary, not the throughput path.
struct freedv {
    int mode = 0, M = 2, Rs = 0, Fs = 0, P = 0;
    int rx_status = 0, verbose = 0, test_frames = 0, frames_per_burst = 0;
    struct FSK *fsk = nullptr;
    pirip_hip_ldpc *ldpc = nullptr;
    pirip_ldpc_info li{};
    LdpcCode code;                              // host copy: Tx framer, frame sizes, the test-frame payload
    std::string code_path;
    std::vector<uint8_t> tf_bytes;
    void *d_in = nullptr; size_t d_in_bytes = 0;
    uint8_t *d_status = nullptr, *d_payload = nullptr; int32_t *d_info = nullptr, *d_nfr = nullptr; int64_t *d_cons = nullptr; float *d_stats = nullptr;
    long frame_periods = 0, period_bits = 0, period_rem = 0;   // freedv_set_verbose: upstream's cycling bit counter (rtl_fsk.cpp does the same)
};

namespace {
bool exists(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
std::string find_code(const char *name)
{
    if (!name || !*name) return "";
    if (exists(name)) return name;
    if (const char *d = getenv("PIRIP_CODE_DIR")) { const std::string p = std::string(d) + "/" + name + ".code"; if (exists(p)) return p; }
    Dl_info di;
    if (dladdr((const void *)&find_code, &di) && di.dli_fname) {
        std::string lib = di.dli_fname;
        const size_t sl = lib.rfind('/');
        const std::string p = (sl == std::string::npos ? std::string(".") : lib.substr(0, sl)) + "/../data/" + name + ".code";
        if (exists(p)) return p;
    }
    return "";
}
void free_dev(struct freedv *f)
{
    void *ptrs[] = {f->d_in, f->d_status, f->d_payload, f->d_info, f->d_nfr, f->d_cons, f->d_stats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
}
}  // namespace

extern "C" {

int pirip_hip_find_code(const char *codename, char *buf, size_t n)
{
    const std::string p = find_code(codename);
    if (p.empty() || !buf || p.size() + 1 > n) return 0;
    memcpy(buf, p.c_str(), p.size() + 1);
    return 1;
}

struct freedv *freedv_open_advanced(int mode, struct freedv_advanced *adv)
{
    if (mode != FREEDV_MODE_FSK_LDPC || !adv) { fprintf(stderr, "libpirip_hip (freedv shim): only FREEDV_MODE_FSK_LDPC is served\n"); return nullptr; }
    if ((adv->M != 2 && adv->M != 4) || adv->Rs <= 0 || adv->Fs <= 0 || adv->Fs % adv->Rs) {
        fprintf(stderr, "libpirip_hip (freedv shim): bad M / Rs / Fs (%d / %d / %d)\n", adv->M, adv->Rs, adv->Fs); return nullptr;
    }
    struct freedv *f = new struct freedv();
    f->mode = mode; f->M = adv->M; f->Rs = adv->Rs; f->Fs = adv->Fs;
    f->code_path = find_code(adv->codename);
    if (f->code_path.empty()) {
        fprintf(stderr, "libpirip_hip (freedv shim): no table for code %s: codec2's LDPC tables are not part of this build; drop %s.code "
                        "(format: pirip_amd/csrc/fsk_ldpc.hpp) into $PIRIP_CODE_DIR or pass a file path as the codename\n",
                adv->codename ? adv->codename : "(null)", adv->codename ? adv->codename : "NAME");
        delete f; return nullptr;
    }
    const std::string err = f->code.load(f->code_path);
    if (!err.empty()) { fprintf(stderr, "libpirip_hip (freedv shim): %s: %s\n", f->code_path.c_str(), err.c_str()); delete f; return nullptr; }
    // "for FSK_LDPC we want the smallest P possible": Ts halved while > 10 and even [UPSTREAM-RECALLED freedv_fsk.c; SURVEY.md 8 table]
    int P = adv->Fs / adv->Rs;
    while (P > 10 && (P % 2) == 0) P /= 2;
    if (P < 4) P = adv->Fs / adv->Rs;
    f->P = P;
    // first_tone / tone_spacing matter to a modulator and to the mask estimator only; rpitx_fsk passes them uninitialised
    const int spacing = (adv->tone_spacing > 0 && adv->tone_spacing < adv->Fs) ? adv->tone_spacing : adv->Rs;
    const int first = (adv->first_tone > 0 && adv->first_tone < adv->Fs) ? adv->first_tone : adv->Rs;
    f->fsk = fsk_create_hbr(adv->Fs, adv->Rs, adv->M, P, PIRIP_FSK_DEFAULT_NSYM, first, spacing);
    fsk_set_freq_est_limits(f->fsk, 0, adv->Fs / 2);      // [UPSTREAM-RECALLED freedv_fsk_ldpc_open]; callers narrow it through freedv_get_fsk()
    std::vector<uint8_t> bits((size_t)f->code.k);
    testframe_payload(bits.data(), f->code.k);
    f->tf_bytes.resize((size_t)f->code.data_bytes());
    pack_bits_msb(f->tf_bytes.data(), bits.data(), f->code.k);
    return f;                                             // the device side is created on the first receive call: a Tx-only program never needs a GPU
}

void freedv_close(struct freedv *f)
{
    if (!f) return;
    if (f->ldpc) pirip_hip_ldpc_destroy(f->ldpc);
    free_dev(f);
    if (f->fsk) fsk_destroy(f->fsk);
    delete f;
}

int freedv_nin(struct freedv *f) { return (int)fsk_nin(f->fsk); }
int freedv_get_n_max_modem_samples(struct freedv *f) { return f->fsk->N + f->fsk->Ts; }
int freedv_get_rx_status(struct freedv *f) { return f->rx_status; }
int freedv_get_bits_per_modem_frame(struct freedv *f) { return f->code.k; }
void freedv_set_frames_per_burst(struct freedv *f, int n) { f->frames_per_burst = n; }
void freedv_set_verbose(struct freedv *f, int v) { f->verbose = v; }
void freedv_set_test_frames(struct freedv *f, int t) { f->test_frames = t; }
struct FSK *freedv_get_fsk(struct freedv *f) { return f->fsk; }
// [UPSTREAM-RECALLED freedv_api.c] the statistics calls a receiver's status display uses: sync = the receiver holds frame sync,
// snr_est = the demodulator's smoothed Eb/N0 figure; the extended form is fsk_get_demod_stats() with sync filled in
int freedv_get_sync(struct freedv *f) { return (f->rx_status & FREEDV_RX_SYNC) ? 1 : 0; }
void freedv_get_modem_stats(struct freedv *f, int *sync, float *snr_est)
{
    if (sync) *sync = freedv_get_sync(f);
    if (snr_est) *snr_est = f->fsk->stats ? f->fsk->stats->snr_est : 0.f;
}
void freedv_get_modem_extended_stats(struct freedv *f, struct MODEM_STATS *stats)
{
    fsk_get_demod_stats(f->fsk, stats);
    stats->sync = freedv_get_sync(f);
}

int freedv_rawdatacomprx(struct freedv *f, unsigned char *packed_payload_bits, COMP demod_in[])
{
    struct FSK *fsk = f->fsk;
    ensure_device(fsk);
    Priv *p = P(fsk);
    int rc;
    if (!f->ldpc) {
        rc = pirip_hip_ldpc_create(f->code_path.c_str(), f->M, fsk->Nsym, 1, -1, &f->ldpc);
        if (rc != PIRIP_OK) die("pirip_hip_ldpc_create", rc);
        pirip_hip_ldpc_get_info(f->ldpc, &f->li);
        bool ok = hipMalloc((void **)&f->d_status, 16) == hipSuccess && hipMalloc((void **)&f->d_payload, (size_t)f->li.data_bytes + 16) == hipSuccess &&
                  hipMalloc((void **)&f->d_info, sizeof(int32_t) * PIRIP_LDPC_INFO_PER_CALL) == hipSuccess && hipMalloc((void **)&f->d_nfr, 16) == hipSuccess &&
                  hipMalloc((void **)&f->d_cons, 16) == hipSuccess && hipMalloc((void **)&f->d_stats, sizeof(float) * PIRIP_STATS_PER_FRAME) == hipSuccess;
        if (!ok) die("hipMalloc", PIRIP_ERR_NOMEM);
    }
    const int nin = fsk->nin;
    const size_t bytes = sizeof(COMP) * (size_t)nin;
    if (bytes > f->d_in_bytes) {
        if (f->d_in) (void)hipFree(f->d_in);
        f->d_in = nullptr; f->d_in_bytes = 0;
        if (hipMalloc(&f->d_in, sizeof(COMP) * (size_t)freedv_get_n_max_modem_samples(f) + 64) != hipSuccess) die("hipMalloc", PIRIP_ERR_NOMEM);
        f->d_in_bytes = sizeof(COMP) * (size_t)freedv_get_n_max_modem_samples(f);
    }
    if (hipMemcpy(f->d_in, demod_in, bytes, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy", PIRIP_ERR_HIP);
    rc = pirip_hip_fsk_ldpc_rx_batch(p->dev, f->ldpc, f->d_in, 0, nin, f->d_status, f->d_payload, f->d_info, f->d_stats, 0, f->d_nfr, f->d_cons, 1, nullptr);
    if (rc != PIRIP_OK) die("pirip_hip_fsk_ldpc_rx_batch", rc);
    if (hipDeviceSynchronize() != hipSuccess) die("hipDeviceSynchronize", PIRIP_ERR_HIP);
    uint8_t st = 0; int32_t nf = 0; int32_t in[PIRIP_LDPC_INFO_PER_CALL] = {0};
    float stats[PIRIP_STATS_PER_FRAME] = {0};
    std::vector<uint8_t> pl((size_t)f->li.data_bytes);
    bool ok = hipMemcpy(&nf, f->d_nfr, sizeof(nf), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&st, f->d_status, 1, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(pl.data(), f->d_payload, pl.size(), hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(in, f->d_info, sizeof(in), hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(stats, f->d_stats, sizeof(stats), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) die("hipMemcpy", PIRIP_ERR_HIP);
    if (nf == 1) p->ran = true;
    refresh(fsk, stats);                                  // nin, f_est, timing, SNRest ... of struct FSK
    f->rx_status = nf == 1 ? st : 0;
    int nbytes = 0;
    if (nf == 1 && (st & FREEDV_RX_BITS)) { memcpy(packed_payload_bits, pl.data(), pl.size()); nbytes = (int)pl.size(); }
    if (nf == 1) {
        f->period_bits += fsk->Nbits;
        if (f->period_bits >= f->li.bits_per_frame) { f->period_bits -= f->li.bits_per_frame; f->period_rem = f->period_bits; f->frame_periods++; }
        if (f->verbose >= 2 && in[6] >= 0) {              // [REF README.md:200-208] one line per decoded frame
            int ecdd = 0;
            if (f->test_frames) for (int b = 2; b < f->li.data_bytes - 2; b++) ecdd += __builtin_popcount((unsigned)(pl[(size_t)b] ^ f->tf_bytes[(size_t)b]));
            const char rxst[5] = {(st & FREEDV_RX_BIT_ERRORS) ? 'E' : '-', (st & FREEDV_RX_BITS) ? 'B' : '-', (st & FREEDV_RX_SYNC) ? 'S' : '-',
                                  (st & FREEDV_RX_TRIAL_SYNC) ? 'T' : '-', 0};
            const double snrdB = 10.0 * log10((double)stats[5] * (double)f->Rs / 3000.0 + 1e-12);
            const int uw_loc = (int)((in[1] + f->period_bits) % f->li.bits_per_frame);
            fprintf(stderr, "%3ld nbits: %3ld state: %d uw_loc: %3d uw_err: %2d bad_uw: %d snrdB: %4.1f eraw: %3d ecdd: %3d iter: %3d pcc: %3d rxst: %s\n",
                    f->frame_periods, f->period_rem, in[0], uw_loc, in[2], in[3], snrdB, in[8], ecdd, in[4], in[5], rxst);
        }
    }
    return nbytes;
}

// Tx side (CPU): what /root/reference/tx/rpitx_fsk.cpp:33-40 declares by hand
int freedv_tx_fsk_ldpc_bits_per_frame(struct freedv *f) { return f->code.bits_per_frame(); }
void freedv_tx_fsk_ldpc_framer(struct freedv *f, uint8_t frame[], uint8_t payload_data[]) { frame_bits(f->code, payload_data, frame); }
unsigned short freedv_gen_crc16(unsigned char *data_p, int length) { return crc16_ccitt(data_p, length); }
void freedv_pack(unsigned char *bytes, unsigned char *bits, int nbits) { pack_bits_msb(bytes, bits, nbits); }
void freedv_unpack(unsigned char *bits, unsigned char *bytes, int nbits) { unpack_bits_msb(bits, bytes, nbits); }
void ofdm_generate_payload_data_bits(uint8_t payload_data_bits[], int n) { testframe_payload(payload_data_bits, n); }

}  // extern "C"
