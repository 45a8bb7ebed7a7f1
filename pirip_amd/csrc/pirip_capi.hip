// pirip_amd/csrc/pirip_capi.hip -- implementation of include/pirip_hip.h sections A and misc.
// Host C++ over the HIP runtime; owns device memory for tables and per-stream state.
// There is deliberately no CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pirip_hip.h"
#include "fsk_device.hpp"
#include "fsk_plan.hpp"
#include "demod_handle.hpp"

using namespace pirip;

#define HIPCHK(expr)                                              \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) {                                   \
            h->last_hip = (int)e_;                                \
            return PIRIP_ERR_HIP;                                 \
        }                                                         \
    } while (0)


namespace {

int bytes_per_sample(int fmt)
{
    switch (fmt) {
    case PIRIP_IN_CU8_FSKDEMOD: case PIRIP_IN_CU8_CSDR: return 2;
    case PIRIP_IN_CS16: return 4;
    default: return 8;
    }
}

template <typename T>
hipError_t upload(T **dst, const void *src, size_t bytes)
{
    hipError_t e = hipMalloc((void **)dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return e;
}

void free_all(pirip_hip_demod *h)
{
    void *ptrs[] = {h->d_hann, h->d_tw, h->d_perm, h->d_lut, h->d_tph, h->d_teeth, h->d_mask_dtheta,
                    h->d_osc_drift, h->d_osc_step, h->d_timing_rec, h->d_fast_tab,
                    h->d_Sf, h->d_theta, h->d_hist, h->d_scal, h->d_phic, h->d_first, h->d_stage_in, h->d_stage_bits,
                    h->d_stage_filt, h->d_stage_stats, h->d_stage_nframes, h->d_stage_consumed, h->d_eye};
    for (void *p : ptrs) if (p) (void)hipFree(p);
}

int reset_state(pirip_hip_demod *h, hipStream_t st)
{
    const FskDims &d = h->plan.d;
    const size_t ns = (size_t)h->nstreams;
    HIPCHK(hipMemsetAsync(h->d_Sf, 0, sizeof(float) * ns * d.Ndft, st));
    HIPCHK(hipMemsetAsync(h->d_theta, 0, sizeof(uint32_t) * ns * kMaxTones, st));
    HIPCHK(hipMemsetAsync(h->d_hist, 0, sizeof(float2) * ns * d.M * d.hist_len, st));
    HIPCHK(hipMemsetAsync(h->d_phic, 0, sizeof(float2) * ns * kMaxTones, st));
    std::vector<StreamScalars> sc(ns);
    std::memset(sc.data(), 0, sizeof(StreamScalars) * ns);
    for (auto &s : sc) s.nin = d.N;
    HIPCHK(hipMemcpyAsync(h->d_scal, sc.data(), sizeof(StreamScalars) * ns, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // sc goes out of scope
    h->nin0 = d.N;
    h->fresh = true;
    return PIRIP_OK;
}

void fill_args(const pirip_hip_demod *h, DemodArgs *a)
{
    a->d = h->plan.d;
    for (int i = 0; i < kMaxStages; i++) a->stages[i] = h->plan.stages[i];
    a->t = DemodTables{h->d_hann, h->d_tw, h->d_perm, h->d_lut, h->d_tph, h->d_teeth, h->d_mask_dtheta,
                       h->d_osc_drift, h->d_osc_step, h->d_timing_rec, h->d_fast_tab};
    for (int i = 0; i < 18; i++) a->tw_s2[i] = h->plan.tw_s2[i];
    a->s = DemodState{h->d_Sf, h->d_theta, h->d_hist, h->d_scal, h->d_phic};
}

// every entry point runs on the device the handle was created on, whatever the caller's current device is
bool bind(const pirip_hip_demod *h)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == h->device) return true;
    return hipSetDevice(h->device) == hipSuccess;
}

}  // namespace

namespace pirip {
void demod_fill_args(const pirip_hip_demod *h, DemodArgs *a) { fill_args(h, a); }
bool demod_bind(const pirip_hip_demod *h) { return bind(h); }
void capture_release(pirip_hip_demod *h);       // capture.hip: frees the handle's capture work area
}  // namespace pirip

extern "C" {

// 0.2: PIRIP_STATS_PER_FRAME 8 -> 10 and two more floats in pirip_stream_state (round 3) -- a caller built against the 0.1 header
// must not be relinked against this library unchanged: pirip_hip_abi() lets it check sizes at start-up instead of being overrun.
const char *pirip_hip_version(void) { return "pirip_hip 0.2 (gfx950)"; }

int pirip_hip_abi(int *abi_version, int *stats_per_frame, size_t *stream_state_bytes)
{
    if (abi_version) *abi_version = PIRIP_HIP_ABI_VERSION;
    if (stats_per_frame) *stats_per_frame = PIRIP_STATS_PER_FRAME;
    if (stream_state_bytes) *stream_state_bytes = sizeof(pirip_stream_state);
    return PIRIP_OK;
}

int pirip_hip_abi_check(int abi_version, int stats_per_frame, size_t stream_state_bytes)
{
    return abi_version == PIRIP_HIP_ABI_VERSION && stats_per_frame == PIRIP_STATS_PER_FRAME && stream_state_bytes == sizeof(pirip_stream_state);
}

const char *pirip_hip_kernel_source_hash(void) { return pirip::demod_wave_source_hash(); }

const char *pirip_hip_strerror(int status)
{
    switch (status) {
    case PIRIP_OK: return "ok";
    case PIRIP_ERR_BAD_ARG: return "bad argument";
    case PIRIP_ERR_BAD_CONFIG: return "bad modem configuration (codec2 fsk_create would assert)";
    case PIRIP_ERR_NO_DEVICE: return "no usable HIP device";
    case PIRIP_ERR_HIP: return "HIP runtime error";
    case PIRIP_ERR_NOMEM: return "out of memory";
    case PIRIP_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown";
    }
}

int pirip_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pirip_hip_selftest_atan2(const float *d_y, const float *d_x, float *d_out, int n)
{
    if (!d_y || !d_x || !d_out || n < 0) return PIRIP_ERR_BAD_ARG;
    if (pirip_hip_device_count() <= 0) return PIRIP_ERR_NO_DEVICE;
    return selftest_atan2(d_y, d_x, d_out, n) == hipSuccess ? PIRIP_OK : PIRIP_ERR_HIP;
}

int pirip_hip_selftest_sqrt(uint64_t *mismatches)
{
    if (!mismatches) return PIRIP_ERR_BAD_ARG;
    if (pirip_hip_device_count() <= 0) return PIRIP_ERR_NO_DEVICE;
    unsigned long long m = 0;
    if (selftest_sqrt(&m) != hipSuccess) return PIRIP_ERR_HIP;
    *mismatches = m;
    return PIRIP_OK;
}

int pirip_hip_selftest_div(uint64_t *mismatches)
{
    if (!mismatches) return PIRIP_ERR_BAD_ARG;
    if (pirip_hip_device_count() <= 0) return PIRIP_ERR_NO_DEVICE;
    unsigned long long m = 0;
    if (selftest_div(&m) != hipSuccess) return PIRIP_ERR_HIP;
    *mismatches = m;
    return PIRIP_OK;
}

static int est_band_for(pirip_hip_demod *h);
void pirip_hip_recalled_defaults(pirip_fsk_recalled *r) { if (r) recalled_defaults(r); }

int pirip_hip_create(const pirip_fsk_params *p, int nstreams, int device, pirip_hip_demod **out)
{
    return pirip_hip_create_recalled(p, nullptr, nstreams, device, out);
}

int pirip_hip_create_recalled(const pirip_fsk_params *p, const pirip_fsk_recalled *recalled, int nstreams, int device, pirip_hip_demod **out)
{
    if (!p || !out || nstreams <= 0) return PIRIP_ERR_BAD_ARG;
    *out = nullptr;
    pirip_hip_demod *h = new (std::nothrow) pirip_hip_demod();
    if (!h) return PIRIP_ERR_NOMEM;
    pirip_fsk_recalled from_env;
    if (!recalled && getenv("PIRIP_RECALLED")) {            // the drill's switch for programs that call pirip_hip_create (the CLI tools, the shims)
        recalled_defaults(&from_env);
        if (!recalled_from_env(&from_env)) { delete h; return PIRIP_ERR_BAD_CONFIG; }
        recalled = &from_env;
    }
    int rc = h->plan.init(p->Fs, p->Rs, p->M, p->P, p->Nsym, p->est_min, p->est_max,
                          p->freq_est_type, p->tone_spacing, p->in_format, recalled);
    if (rc != PIRIP_OK) { delete h; return rc; }
    if (demod_general_lds_bytes(h->plan.d) > 160 * 1024) { delete h; return PIRIP_ERR_UNSUPPORTED; }

    int ndev = pirip_hip_device_count();
    if (ndev <= 0) { delete h; return PIRIP_ERR_NO_DEVICE; }
    if (device >= 0) { if (device >= ndev || hipSetDevice(device) != hipSuccess) { delete h; return PIRIP_ERR_NO_DEVICE; } }
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return PIRIP_ERR_NO_DEVICE; }
    h->nstreams = nstreams;
    {
        const char *k = getenv("PIRIP_KERNEL");
        const bool want_general = getenv("PIRIP_FORCE_GENERAL") || (k && !strcmp(k, "general"));
        if (const char *f = getenv("PIRIP_FFT_FMA")) {
            // opt-in A/B switch: only where a fused instance was built (the headline shape), else the exact kernel stays
            h->plan.d.fft_fma = atoi(f) ? 1 : 0;
            if (h->plan.d.fft_fma && !demod_wave_applicable(h->plan.d)) h->plan.d.fft_fma = 0;
        }
        // (the specialised instances are built around the recalled constants' defaults: another value of one of them -> the general kernel)
        h->kernel = want_general || !h->plan.d.recalled_fast_ok ? 0 :
                    demod_wave_applicable(h->plan.d) ? PIRIP_KERNEL_WAVE : demod_block_applicable(h->plan.d) ? PIRIP_KERNEL_BLOCK : 0;
        if (const char *sg = getenv("PIRIP_BLOCK_STAGGER")) h->plan.d.block_stagger = atoi(sg) > 0 ? atoi(sg) : 0;
        if (const char *b = getenv("PIRIP_EST_BAND")) {
            // opt-in switch for the command-line tools: the band-only estimator where it applies (include/pirip_hip.h), else nothing
            if (atoi(b)) h->plan.d.est_band = est_band_for(h);
        }
        if (k && !strcmp(k, "exact")) {
            // every frame in the oracle's operation order (fsk_demod_general.hip, EXACT == 2): integrator memory as single samples
            h->plan.d.grp = 1;
            if (!demod_exact_applicable(h->plan.d)) { delete h; return PIRIP_ERR_UNSUPPORTED; }
            h->kernel = PIRIP_KERNEL_EXACT;
        }
    }

    const FskPlan &pl = h->plan;
    const FskDims &d = pl.d;
    const size_t ns = (size_t)nstreams;
    bool ok = true;
    ok &= upload(&h->d_hann, pl.hann.data(), sizeof(float) * d.Ndft) == hipSuccess;
    ok &= upload(&h->d_tw, pl.twiddle.data(), sizeof(float) * 2 * d.Ndft) == hipSuccess;
    ok &= upload(&h->d_perm, pl.leaf_iperm.data(), sizeof(uint16_t) * d.Ndft) == hipSuccess;
    ok &= upload(&h->d_lut, pl.u8_lut.data(), sizeof(float) * 256) == hipSuccess;
    ok &= upload(&h->d_tph, pl.timing_ph.data(), sizeof(float) * 2 * d.P) == hipSuccess;
    ok &= upload(&h->d_teeth, pl.teeth.data(), sizeof(int16_t) * pl.teeth.size()) == hipSuccess;
    ok &= upload(&h->d_mask_dtheta, pl.mask_dtheta.data(), sizeof(uint32_t) * pl.mask_dtheta.size()) == hipSuccess;
    ok &= upload(&h->d_osc_drift, pl.osc_drift.data(), sizeof(float) * pl.osc_drift.size()) == hipSuccess;
    ok &= upload(&h->d_osc_step, pl.osc_step.data(), sizeof(float) * pl.osc_step.size()) == hipSuccess;
    ok &= upload(&h->d_timing_rec, pl.timing_rec.data(), sizeof(float) * pl.timing_rec.size()) == hipSuccess;
    ok &= upload(&h->d_fast_tab, pl.fast_tab.data(), sizeof(float) * pl.fast_tab.size()) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_Sf, sizeof(float) * ns * d.Ndft) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_theta, sizeof(uint32_t) * ns * kMaxTones) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_hist, sizeof(float2) * ns * d.M * d.hist_len) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_scal, sizeof(StreamScalars) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_first, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&h->d_phic, sizeof(float2) * ns * kMaxTones) == hipSuccess;
    if (!ok) { free_all(h); delete h; return PIRIP_ERR_NOMEM; }
    if (const char *e = getenv("PIRIP_EXACT0")) h->exact0 = atoi(e) ? 1 : 0;
    rc = reset_state(h, nullptr);
    if (rc != PIRIP_OK) { free_all(h); delete h; return rc; }
    *out = h;
    return PIRIP_OK;
}

int pirip_hip_destroy(pirip_hip_demod *h)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    (void)bind(h);
    (void)hipDeviceSynchronize();
    pirip::capture_release(h);
    free_all(h);
    delete h;
    return PIRIP_OK;
}

int pirip_hip_get_kernel(const pirip_hip_demod *h) { return h ? h->kernel : PIRIP_ERR_BAD_ARG; }

int pirip_hip_get_kernel_name(const pirip_hip_demod *h, char *buf, size_t n)
{
    if (!h || !buf || !n) return PIRIP_ERR_BAD_ARG;
    buf[0] = 0;
    if (h->kernel == 2 && demod_wave_describe(h->plan.d, buf, n) > 0) return PIRIP_OK;
    if (h->kernel == PIRIP_KERNEL_BLOCK && demod_block_describe(h->plan.d, buf, n) > 0) return PIRIP_OK;
    const FskDims &d = h->plan.d;
    snprintf(buf, n, "fsk_demod_%s_kernel(M=%d,Ts=%d,P=%d,Nsym=%d,Ndft=%d,format %d%s)", h->kernel == PIRIP_KERNEL_EXACT ? "exact" : "general", d.M, d.Ts, d.P, d.Nsym, d.Ndft, d.in_format,
             d.freq_est_type ? ",mask estimator" : "");
    return PIRIP_OK;
}

int pirip_hip_clear_estimators(pirip_hip_demod *h, void *hip_stream)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const FskDims &d = h->plan.d;
    const size_t ns = (size_t)h->nstreams;
    HIPCHK(hipMemsetAsync(h->d_Sf, 0, sizeof(float) * ns * d.Ndft, st));
    // nin = N in every stream's scalars: a strided copy of one int per stream
    std::vector<int32_t> nin(ns, d.N);
    HIPCHK(hipMemcpy2DAsync(&h->d_scal->nin, sizeof(StreamScalars), nin.data(), sizeof(int32_t), sizeof(int32_t), ns, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // nin goes out of scope
    h->nin0 = d.N;
    return PIRIP_OK;
}

int pirip_hip_get_info(const pirip_hip_demod *h, pirip_fsk_info *info)
{
    if (!h || !info) return PIRIP_ERR_BAD_ARG;
    const FskDims &d = h->plan.d;
    info->Ts = d.Ts; info->N = d.N; info->Nmem = d.Nmem; info->Ndft = d.Ndft; info->Nbits = d.Nbits;
    info->nin_max = d.N + d.nin_step; info->nstreams = h->nstreams;
    info->bytes_per_sample = bytes_per_sample(d.in_format);
    return PIRIP_OK;
}

int pirip_hip_reset(pirip_hip_demod *h, void *hip_stream)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    return reset_state(h, (hipStream_t)hip_stream);
}

}  // extern "C"

namespace pirip {
// Before the demodulator proper: when every stream of the handle is still in its created state and this call holds its first frame,
// that frame is demodulated by the exact prologue (same outputs, state left in the handle's kernel's layout) and a->io.first tells the
// launch that follows where each stream goes on. a->io must be complete. Shapes without the prologue: nothing happens.
int exact0_prologue(pirip_hip_demod *h, DemodArgs *a, hipStream_t st)
{
    const FskDims &d = h->plan.d;
    if (!h->fresh) return PIRIP_OK;
    if (a->io.nsamp < d.N || a->io.max_frames < 1) return PIRIP_OK;          // no frame in this call: the streams stay as created
    h->fresh = false;
    if (!h->exact0 || !demod_exact0_applicable(d) || h->kernel == PIRIP_KERNEL_BLOCK || h->kernel == PIRIP_KERNEL_EXACT || a->io.soft.llr || a->io.seg) return PIRIP_OK;
    DemodArgs p = *a;
    p.io.first = nullptr; p.io.first_out = h->d_first;
    p.io.exact0_fmt = h->kernel == PIRIP_KERNEL_WAVE ? PIRIP_KERNEL_WAVE : PIRIP_KERNEL_GENERAL;
    p.io.eye = h->d_eye;
    const hipError_t e = launch_demod_exact0(p, h->nstreams, st);
    if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
    a->io.first = h->d_first;
    return PIRIP_OK;
}
}  // namespace pirip

extern "C" {

int pirip_hip_set_exact_first_frame(pirip_hip_demod *h, int enable)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    h->exact0 = enable ? 1 : 0;
    return PIRIP_OK;
}

int pirip_hip_demod_batch(pirip_hip_demod *h, const void *d_in, size_t in_stride_bytes, int64_t nsamp,
                          uint8_t *d_bits, size_t bits_stride, float *d_rx_filt, size_t filt_stride,
                          float *d_stats, size_t stats_stride, int32_t *d_nframes, int64_t *d_consumed,
                          int64_t max_frames, void *hip_stream)
{
    if (!h || !d_in || nsamp < 0 || max_frames < 0) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    DemodArgs a;
    fill_args(h, &a);
    a.io = DemodIO{(const uint8_t *)d_in, in_stride_bytes, nsamp, d_bits, bits_stride, d_rx_filt, filt_stride,
                   d_stats, stats_stride, d_nframes, d_consumed, max_frames, SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}};
    a.io.eye = (h->kernel == PIRIP_KERNEL_GENERAL || h->kernel == PIRIP_KERNEL_EXACT) ? h->d_eye : nullptr;
    hipError_t e;
    if (h->kernel == 2 && nsamp > demod_wave_max_samples(a.d)) return PIRIP_ERR_UNSUPPORTED;   // present the batch in smaller pieces (before anything runs)
    {
        const int pr = exact0_prologue(h, &a, (hipStream_t)hip_stream);
        if (pr != PIRIP_OK) return pr;
    }
    if (h->kernel == 2) {
        e = launch_demod_wave(a, h->nstreams, (hipStream_t)hip_stream);
    } else e = launch_demod_kind(h->kernel, a, h->nstreams, (hipStream_t)hip_stream);
    if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
    return PIRIP_OK;
}

}  // extern "C"

// internal (ldpc_kernels.hip): one batch with the fused FSK_LDPC hand-over instead of bits / magnitudes
namespace pirip {
// s0 / n (n < 0: all): streams [s0, s0 + n) of the handle only -- every per-stream array of the argument block is advanced to stream s0, the
// pointers the caller passes are those of stream 0 (pirip_hip_fsk_ldpc_rx_batch runs two ranges on two HIP streams)
int demod_batch_soft(pirip_hip_demod *h, const void *d_in, size_t in_stride_bytes, int64_t nsamp, const SoftOut &so, float *d_stats, size_t stats_stride,
                     int32_t *d_nframes, int64_t *d_consumed, int64_t max_frames, hipStream_t st, int s0, int n)
{
    if (!h || !d_in || nsamp < 0 || max_frames < 0 || !so.llr || !so.words || !so.lnI0 || (so.bit0 & 31)) return PIRIP_ERR_BAD_ARG;
    if (h->kernel != 2 || !demod_wave_soft_capable(h->plan.d) || nsamp > demod_wave_max_samples(h->plan.d)) return PIRIP_ERR_UNSUPPORTED;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    DemodArgs a;
    fill_args(h, &a);
    a.io = DemodIO{(const uint8_t *)d_in, in_stride_bytes, nsamp, nullptr, 0, nullptr, 0, d_stats, stats_stride, d_nframes, d_consumed, max_frames, so};
    (void)exact0_prologue(h, &a, st);                    // (the fused hand-over has no prologue: this only notes that the streams have started)
    int nrun = h->nstreams;
    if (n >= 0) {
        if (s0 < 0 || s0 + n > h->nstreams) return PIRIP_ERR_BAD_ARG;
        const FskDims &d = h->plan.d;
        const size_t z = (size_t)s0;
        a.s.Sf += z * d.Ndft; a.s.theta += z * kMaxTones; a.s.hist += z * d.M * d.hist_len; a.s.scal += z;
        if (a.s.phic) a.s.phic += z * kMaxTones;
        a.io.in += z * in_stride_bytes;
        if (a.io.stats) a.io.stats += z * stats_stride;
        a.io.nframes += z;
        if (a.io.consumed) a.io.consumed += z;
        a.io.soft.llr += z * so.llr_stride; a.io.soft.words += z * so.words_stride;
        nrun = n;
    }
    if (nrun == 0) return PIRIP_OK;
    const hipError_t e = launch_demod_wave(a, nrun, st);
    if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
    return PIRIP_OK;
}
bool demod_soft_capable(const pirip_hip_demod *h, int64_t nsamp)
{
    return h && h->kernel == 2 && demod_wave_soft_capable(h->plan.d) && nsamp <= demod_wave_max_samples(h->plan.d);
}
int demod_streams_per_cu(const pirip_hip_demod *h) { return h && h->kernel == 2 ? demod_wave_streams_per_cu(h->plan.d) : 0; }
int demod_handle_shape(const pirip_hip_demod *h, int *M, int *Nsym, int *nstreams, int *device)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    *M = h->plan.d.M; *Nsym = h->plan.d.Nsym; *nstreams = h->nstreams; *device = h->device;
    return PIRIP_OK;
}
}  // namespace pirip

extern "C" {

int pirip_hip_demod_host(pirip_hip_demod *h, const void *in, int64_t nsamp, uint8_t *bits, float *rx_filt,
                         float *stats, int64_t max_frames, int64_t *nframes, int64_t *consumed)
{
    if (!h || (!in && nsamp > 0) || nsamp < 0 || max_frames < 0) return PIRIP_ERR_BAD_ARG;
    if (h->nstreams != 1) return PIRIP_ERR_BAD_ARG;      // one staged buffer, one set of outputs: a one-stream convenience
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    const FskDims &d = h->plan.d;
    const size_t bps = (size_t)bytes_per_sample(d.in_format);
    const size_t in_bytes = (size_t)nsamp * bps;
    if (in_bytes > h->stage_in_bytes) {
        if (h->d_stage_in) (void)hipFree(h->d_stage_in);
        h->d_stage_in = nullptr; h->stage_in_bytes = 0;
        HIPCHK(hipMalloc(&h->d_stage_in, in_bytes + 64));
        h->stage_in_bytes = in_bytes;
    }
    if (!h->d_stage_in) { HIPCHK(hipMalloc(&h->d_stage_in, 64)); h->stage_in_bytes = 0; }
    if (max_frames > h->stage_frames) {
        void *olds[] = {h->d_stage_bits, h->d_stage_filt, h->d_stage_stats};
        for (void *p : olds) if (p) (void)hipFree(p);
        h->d_stage_bits = nullptr; h->d_stage_filt = nullptr; h->d_stage_stats = nullptr; h->stage_frames = 0;
        HIPCHK(hipMalloc((void **)&h->d_stage_bits, (size_t)max_frames * d.Nbits + 16));
        HIPCHK(hipMalloc((void **)&h->d_stage_filt, sizeof(float) * (size_t)max_frames * d.M * d.Nsym + 16));
        HIPCHK(hipMalloc((void **)&h->d_stage_stats, sizeof(float) * (size_t)max_frames * PIRIP_STATS_PER_FRAME + 16));
        h->stage_frames = max_frames;
    }
    if (!h->d_stage_nframes) {
        HIPCHK(hipMalloc((void **)&h->d_stage_nframes, sizeof(int32_t) * (size_t)h->nstreams));
        HIPCHK(hipMalloc((void **)&h->d_stage_consumed, sizeof(int64_t) * (size_t)h->nstreams));
    }
    if (in_bytes) HIPCHK(hipMemcpy(h->d_stage_in, in, in_bytes, hipMemcpyHostToDevice));
    // all streams of the handle see the same staged buffer (stride 0); callers of this
    // convenience entry normally create the handle with nstreams == 1
    int rc = pirip_hip_demod_batch(h, h->d_stage_in, 0, nsamp, h->d_stage_bits, 0,
                                   rx_filt ? h->d_stage_filt : nullptr, 0,
                                   h->d_stage_stats, 0, h->d_stage_nframes, h->d_stage_consumed,
                                   max_frames, nullptr);
    if (rc != PIRIP_OK) return rc;
    HIPCHK(hipDeviceSynchronize());
    int32_t nf = 0; int64_t cons = 0;
    HIPCHK(hipMemcpy(&nf, h->d_stage_nframes, sizeof(nf), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&cons, h->d_stage_consumed, sizeof(cons), hipMemcpyDeviceToHost));
    const size_t fb = d.pack_bits ? (size_t)(d.Nbits + 7) / 8 : (size_t)d.Nbits;
    if (bits && nf) HIPCHK(hipMemcpy(bits, h->d_stage_bits, (size_t)nf * fb, hipMemcpyDeviceToHost));
    if (rx_filt && nf) HIPCHK(hipMemcpy(rx_filt, h->d_stage_filt, sizeof(float) * (size_t)nf * d.M * d.Nsym, hipMemcpyDeviceToHost));
    if (stats && nf) HIPCHK(hipMemcpy(stats, h->d_stage_stats, sizeof(float) * (size_t)nf * PIRIP_STATS_PER_FRAME, hipMemcpyDeviceToHost));
    StreamScalars sc;
    HIPCHK(hipMemcpy(&sc, h->d_scal, sizeof(sc), hipMemcpyDeviceToHost));
    h->nin0 = sc.nin;
    if (nframes) *nframes = nf;
    if (consumed) *consumed = cons;
    return PIRIP_OK;
}

int pirip_hip_nin0(pirip_hip_demod *h) { return h ? h->nin0 : PIRIP_ERR_BAD_ARG; }

int pirip_hip_get_Sf(pirip_hip_demod *h, int s, float *Sf_host)
{
    if (!h || !Sf_host || s < 0 || s >= h->nstreams) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(Sf_host, h->d_Sf + (size_t)s * h->plan.d.Ndft, sizeof(float) * h->plan.d.Ndft, hipMemcpyDeviceToHost));
    return PIRIP_OK;
}

// MODEM_STATS.rx_eye (include/pirip_hip.h): only the any-configuration kernel keeps every integrator position of a frame in LDS, so
// asking for the eye moves the handle to it; the kernels keep the integrator memory in different layouts, hence the state reset.
int pirip_hip_enable_eye(pirip_hip_demod *h, int enable)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    if (!enable) {
        if (h->d_eye) (void)hipFree(h->d_eye);
        h->d_eye = nullptr;
        return PIRIP_OK;                       // the handle stays on the kernel it has
    }
    if (h->d_eye) return PIRIP_OK;
    const size_t bytes = sizeof(float) * (size_t)h->nstreams * kEyeTraces * kEyePoints;
    HIPCHK(hipMalloc((void **)&h->d_eye, bytes));
    HIPCHK(hipMemset(h->d_eye, 0, bytes));
    if (h->kernel != PIRIP_KERNEL_GENERAL && h->kernel != PIRIP_KERNEL_EXACT) {
        h->kernel = PIRIP_KERNEL_GENERAL;
        return reset_state(h, nullptr);
    }
    return PIRIP_OK;
}

int pirip_hip_get_eye(pirip_hip_demod *h, int s, int normalise, float *rx_eye, int *neyetr, int *neyesamp)
{
    if (!h || !rx_eye || !neyetr || !neyesamp || s < 0 || s >= h->nstreams) return PIRIP_ERR_BAD_ARG;
    if (!h->d_eye) return PIRIP_ERR_UNSUPPORTED;      // pirip_hip_enable_eye() first
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(rx_eye, h->d_eye + (size_t)s * kEyeTraces * kEyePoints, sizeof(float) * kEyeTraces * kEyePoints, hipMemcpyDeviceToHost));
    const FskDims &d = h->plan.d;
    const int dec = (2 * d.P + kEyePoints - 1) / kEyePoints, npts = (2 * d.P) / dec;
    int traces = kEyeTraces / d.M;
    while (traces > 0 && 2 * d.P * traces + (d.P / 2 + 1) + (npts - 1) * dec >= (d.Nsym + 1) * d.P) traces--;   // as the kernel counts them
    if (normalise) {   // [UPSTREAM-RECALLED fsk.c]: every trace divided by the largest value of all of them
        float eye_max = 0.f;
        for (int i = 0; i < traces * d.M; i++)
            for (int j = 0; j < npts; j++) if (fabsf(rx_eye[i * kEyePoints + j]) > eye_max) eye_max = fabsf(rx_eye[i * kEyePoints + j]);
        if (eye_max > 0.f)                     // no frame yet (or a silent one): the zero traces stay zero, not 0/0
            for (int i = 0; i < traces * d.M; i++)
                for (int j = 0; j < npts; j++) rx_eye[i * kEyePoints + j] = rx_eye[i * kEyePoints + j] / eye_max;
    }
    *neyetr = traces * d.M; *neyesamp = npts;
    return PIRIP_OK;
}

// output format of d_bits: 0 = one byte per bit (the reference's stdout format), 1 = 8 bits per byte, MSB first
int pirip_hip_set_bit_packing(pirip_hip_demod *h, int packed)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    h->plan.d.pack_bits = packed ? 1 : 0;
    return PIRIP_OK;
}

// Band-only estimator (opt-in): see include/pirip_hip.h. Eligible where the peak search reads FFT bins 0..31 only and a wave instance
// was built for it; switching it OFF resets the streams (Sf outside the band was not maintained while it was on).
// the narrowest band (in 16-bin columns of the Ndft = 256 dataflow: 2 or 4) that holds the handle's search range and has an instance; 0: none
static int est_band_for(pirip_hip_demod *h)
{
    pirip::FskDims d = h->plan.d;
    if (h->kernel != PIRIP_KERNEL_WAVE || d.freq_est_type != 0 || d.Ndft != 256 || d.est_st < d.Ndft / 2) return 0;
    for (int b = 2; b <= 4; b += 2) {
        d.est_band = b;
        if (d.est_en <= d.Ndft / 2 + 16 * b && demod_wave_applicable(d)) return b;
    }
    return 0;
}
int pirip_hip_set_estimator_band_only(pirip_hip_demod *h, int enable)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    if (enable) {
        const int b = est_band_for(h);
        if (!b) return PIRIP_ERR_UNSUPPORTED;
        h->plan.d.est_band = b;
        return PIRIP_OK;
    }
    if (!h->plan.d.est_band) return PIRIP_OK;
    h->plan.d.est_band = 0;
    return pirip_hip_reset(h, nullptr);
}

// fsk_enable_burst_mode() for every stream of the handle: nin is pinned to N from now on
int pirip_hip_set_burst_mode(pirip_hip_demod *h, int enable)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    h->plan.d.burst_mode = enable ? 1 : 0;
    if (enable) {
        if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
        HIPCHK(hipDeviceSynchronize());
        std::vector<StreamScalars> sc((size_t)h->nstreams);
        HIPCHK(hipMemcpy(sc.data(), h->d_scal, sizeof(StreamScalars) * sc.size(), hipMemcpyDeviceToHost));
        for (auto &s : sc) s.nin = h->plan.d.N;
        HIPCHK(hipMemcpy(h->d_scal, sc.data(), sizeof(StreamScalars) * sc.size(), hipMemcpyHostToDevice));
        h->nin0 = h->plan.d.N;
    }
    return PIRIP_OK;
}

// scalar state of stream s (used by the codec2 shim and tests)
int pirip_hip_get_scalars(pirip_hip_demod *h, int s, float *out8)
{
    if (!h || !out8 || s < 0 || s >= h->nstreams) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    StreamScalars sc;
    HIPCHK(hipMemcpy(&sc, h->d_scal + s, sizeof(sc), hipMemcpyDeviceToHost));
    out8[0] = sc.f_est[0]; out8[1] = sc.f_est[1]; out8[2] = sc.f_est[2]; out8[3] = sc.f_est[3];
    out8[4] = sc.norm_rx_timing; out8[5] = sc.SNRest; out8[6] = (float)sc.nin; out8[7] = sc.ppm;
    return PIRIP_OK;
}

int pirip_hip_get_stream_state(pirip_hip_demod *h, int s, pirip_stream_state *out)
{
    if (!h || !out || s < 0 || s >= h->nstreams) return PIRIP_ERR_BAD_ARG;
    if (!bind(h)) return PIRIP_ERR_NO_DEVICE;
    HIPCHK(hipDeviceSynchronize());
    StreamScalars sc;
    HIPCHK(hipMemcpy(&sc, h->d_scal + s, sizeof(sc), hipMemcpyDeviceToHost));
    out->nin = sc.nin; out->norm_rx_timing = sc.norm_rx_timing; out->ppm = sc.ppm; out->snr_est = sc.snr_est;
    out->SNRest = sc.SNRest; out->EbNodB = sc.EbNodB; out->v_est = sc.v_est;
    for (int m = 0; m < 4; m++) out->f_est[m] = sc.f_est[m];
    out->rx_sig_pow = sc.rx_sig_pow; out->rx_nse_pow = sc.rx_nse_pow;
    return PIRIP_OK;
}

// fsk_set_freq_est_limits() in place: only the peak-search bin range changes (kernel argument), all stream state stays
int pirip_hip_set_freq_est_limits(pirip_hip_demod *h, int est_min, int est_max)
{
    if (!h) return PIRIP_ERR_BAD_ARG;
    int st, en;
    if (!fsk_est_range(h->plan.d.Fs, h->plan.d.Ndft, est_min, est_max, &st, &en)) return PIRIP_ERR_BAD_CONFIG;
    // (a band-only estimator has no history outside its band: the range can move inside it, not out of it)
    if (h->plan.d.est_band && (st < h->plan.d.Ndft / 2 || en > h->plan.d.Ndft / 2 + 16 * h->plan.d.est_band)) return PIRIP_ERR_UNSUPPORTED;
    h->plan.d.est_st = st; h->plan.d.est_en = en;
    return PIRIP_OK;
}

}  // extern "C"
