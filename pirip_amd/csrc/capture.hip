// pirip_amd/csrc/capture.hip -- pirip_hip_demod_capture: ONE long capture (what `fsk_demod` gets from a file) demodulated on
// many wavefronts, results identical to the sequential read loop.
//
// fsk_demod() is a chain: every frame starts where the last one's timing estimate (nin) said, with the smoothed spectrum Sf and the
// integrator-memory tail the frames before it left. One stream = one wavefront therefore used 1 of the chip's 3072 resident waves on a
// single capture. The chain forgets, though:
//   * Sf is a one-pole average (x0.9 per FFT, 5..8 FFTs per frame): after ~400 FFTs it no longer depends on where it started, bit for bit;
//   * the integrator-memory tail is the previous frame's samples mixed with the previous frame's tones (the wave kernel keeps no
//     oscillator phase across frames: every frame's down-conversion starts at phase 0 and the tail is turned to match);
//   * the timing estimate of a frame has no memory at all.
// So the capture is cut into segments of F frames, one wave each. Segment s first demodulates segment s-1's samples from a COLD state
// (the warm-up: no output; started on the frame grid a short pilot run from the stream's true state settles on, its first frames with nin
// pinned: a cold start's first timing estimates are not to be acted on), then -- its state snapshotted -- its own F frames. Afterwards
// the end state of segment s-1 is compared with the snapshot of segment s: Sf, tail, nin, timing and the sample position, bit for bit.
// Equal state + same samples = same results, so a segment whose predecessor is verified and whose snapshot matches is verified.
// Segment 0 starts from the handle's true state. Where a comparison fails -- in practice: the timing loop slipped a whole symbol inside
// a segment and the cold warm-up over the same samples went round the other way, or a sample-clock offset has moved the frame grid away
// from the guessed first samples -- the first failing segment is re-run from its predecessor's verified end state and every later one
// is speculated again, first samples from the prefix sums of the lengths the segments themselves measured. Each pass verifies at least
// one more segment, so the worst case is the sequential loop's cost (x2 for the warm-ups); the usual case is one pass, or a few.
//
// Exact, not approximate: nothing is accepted on a tolerance. The one value of the per-frame statistics that the comparison does not
// cover -- ppm, a one-pole average (x0.9 per frame) of the timing differences, which feeds nothing else -- is recomputed over the
// whole capture from the verified timing column, in frame order, with the demodulator's own expression.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pirip_hip.h"
#include "demod_handle.hpp"

using namespace pirip;

struct CaptureWork {
    int slots = 0;
    // snapshot of every slot's state after its warm-up (same layouts as the handle's state arrays)
    float *d_Sf = nullptr; uint32_t *d_theta = nullptr; float2 *d_hist = nullptr; StreamScalars *d_scal = nullptr;
    SegDesc *d_segA = nullptr, *d_segA2 = nullptr, *d_segB = nullptr;
    int64_t *d_consA = nullptr, *d_consB = nullptr, *d_posB = nullptr;
    int32_t *d_nfA = nullptr, *d_nfB = nullptr, *d_ok = nullptr, *d_mode = nullptr;
    float *d_stats = nullptr; size_t stats_rows = 0;
    float *d_warm_stats = nullptr; size_t warm_rows = 0;   // statistics rows of the warm-up frames (never read: they make every warm-up frame
                                                           // an observable one, so that snr_est's average runs through them)
    StreamScalars *d_scal0 = nullptr;            // the stream's scalars at entry (ppm / timing the recomputation starts from)
    // staging of the host-buffer form
    void *d_in = nullptr; size_t in_bytes = 0;
    uint8_t *d_bits = nullptr; float *d_filt = nullptr; float *d_ostats = nullptr; int64_t out_frames = 0;
};

namespace {

#define CAPCHK(expr)                                              \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) {                                   \
            h->last_hip = (int)e_;                                \
            return PIRIP_ERR_HIP;                                 \
        }                                                         \
    } while (0)

constexpr int kThreads = 256;

// cold state (what fsk_create leaves) for every slot that warms up in this pass
__global__ void cold_kernel(DemodState st, const SegDesc *segA, int Ndft, int hist_elems, int N)
{
    const int s = blockIdx.x;
    if (segA[s].max_frames < 0) return;
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) st.Sf[(size_t)s * Ndft + i] = 0.0f;
    for (int i = threadIdx.x; i < hist_elems; i += blockDim.x) st.hist[(size_t)s * hist_elems + i] = make_float2(0.f, 0.f);
    if (threadIdx.x < kMaxTones) st.theta[(size_t)s * kMaxTones + threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        StreamScalars sc;
        memset(&sc, 0, sizeof(sc));
        sc.nin = N;
        st.scal[s] = sc;
    }
}

// dst slot <- src slot (the true end state of the predecessor becomes the start state of the segment that is re-run exactly)
__global__ void copy_state_kernel(DemodState st, int dst, int src, int Ndft, int hist_elems)
{
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) st.Sf[(size_t)dst * Ndft + i] = st.Sf[(size_t)src * Ndft + i];
    for (int i = threadIdx.x; i < hist_elems; i += blockDim.x) st.hist[(size_t)dst * hist_elems + i] = st.hist[(size_t)src * hist_elems + i];
    if (threadIdx.x < kMaxTones) st.theta[(size_t)dst * kMaxTones + threadIdx.x] = st.theta[(size_t)src * kMaxTones + threadIdx.x];
    if (threadIdx.x == 0) st.scal[dst] = st.scal[src];
}

// after the warm-up launch: snapshot the warmed-up slots, and turn the warm-up descriptors into the segments' own
__global__ void after_warmup_kernel(DemodState st, DemodState snap, const SegDesc *segA, SegDesc *segB, const int64_t *consA, int64_t *posB,
                                    int Ndft, int hist_elems, int v, int64_t true_pos)
{
    const int s = blockIdx.x;
    if (s == v && threadIdx.x == 0) posB[s] = true_pos;         // the segment that starts from the true state: no warm-up, exact position
    if (segA[s].max_frames < 0) return;
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) snap.Sf[(size_t)s * Ndft + i] = st.Sf[(size_t)s * Ndft + i];
    for (int i = threadIdx.x; i < hist_elems; i += blockDim.x) snap.hist[(size_t)s * hist_elems + i] = st.hist[(size_t)s * hist_elems + i];
    if (threadIdx.x < kMaxTones) snap.theta[(size_t)s * kMaxTones + threadIdx.x] = st.theta[(size_t)s * kMaxTones + threadIdx.x];
    if (threadIdx.x == 0) {
        snap.scal[s] = st.scal[s];
        const int64_t p = segA[s].in_off + consA[s];
        segB[s].in_off = p;
        posB[s] = p;
    }
}

// the segment at the head of the unverified part: its start state is the (verified) end state of its predecessor, its first sample the
// predecessor's last + 1. The predecessor does not run in this pass.
__global__ void continue_kernel(DemodState st, DemodState snap, const int32_t *mode, SegDesc *segB, const int64_t *consB, int64_t *posB, int Ndft,
                                int hist_elems)
{
    const int s = blockIdx.x;
    if (mode[s] != 2 || s == 0) return;
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) {
        const float x = st.Sf[(size_t)(s - 1) * Ndft + i];
        st.Sf[(size_t)s * Ndft + i] = x; snap.Sf[(size_t)s * Ndft + i] = x;
    }
    for (int i = threadIdx.x; i < hist_elems; i += blockDim.x) {
        const float2 x = st.hist[(size_t)(s - 1) * hist_elems + i];
        st.hist[(size_t)s * hist_elems + i] = x; snap.hist[(size_t)s * hist_elems + i] = x;
    }
    if (threadIdx.x < kMaxTones) {
        const uint32_t x = st.theta[(size_t)(s - 1) * kMaxTones + threadIdx.x];
        st.theta[(size_t)s * kMaxTones + threadIdx.x] = x; snap.theta[(size_t)s * kMaxTones + threadIdx.x] = x;
    }
    if (threadIdx.x == 0) {
        const StreamScalars x = st.scal[s - 1];
        st.scal[s] = x; snap.scal[s] = x;
        const int64_t p = posB[s - 1] + consB[s - 1];
        posB[s] = p;
        segB[s].in_off = p;
    }
}

// ok[s] = 0 when the state segment s started its own frames from is, bit for bit, the state segment s-1 ended in, at the same sample
__global__ void verify_kernel(DemodState st, DemodState snap, const int64_t *posB, const int64_t *consB, int32_t *ok, int first, int Ndft,
                              int hist_elems, int M)
{
    const int s = first + blockIdx.x;          // compares end of s-1 with snapshot of s
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int b = 0;                                 // what differs: 1 Sf, 2 integrator tail, 4 oscillator phase, 8 nin, 16 timing, 32 sample position
    const uint32_t *a0 = (const uint32_t *)(st.Sf + (size_t)(s - 1) * Ndft), *b0 = (const uint32_t *)(snap.Sf + (size_t)s * Ndft);
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) b |= a0[i] != b0[i] ? 1 : 0;
    const uint32_t *a1 = (const uint32_t *)(st.hist + (size_t)(s - 1) * hist_elems), *b1 = (const uint32_t *)(snap.hist + (size_t)s * hist_elems);
    for (int i = threadIdx.x; i < 2 * hist_elems; i += blockDim.x) b |= a1[i] != b1[i] ? 2 : 0;
    if (threadIdx.x < M) b |= st.theta[(size_t)(s - 1) * kMaxTones + threadIdx.x] != snap.theta[(size_t)s * kMaxTones + threadIdx.x] ? 4 : 0;
    if (threadIdx.x == 0) {
        const StreamScalars x = st.scal[s - 1], y = snap.scal[s];
        b |= x.nin != y.nin ? 8 : 0;
        b |= __float_as_uint(x.norm_rx_timing) != __float_as_uint(y.norm_rx_timing) ? 16 : 0;
        b |= posB[s - 1] + consB[s - 1] != posB[s] ? 32 : 0;
    }
    if (b) atomicOr(&bad, b);
    __syncthreads();
    if (threadIdx.x == 0) ok[s] = bad;          // 0 = verified
}

// ppm over the whole capture, in frame order, with the demodulator's own expression (fsk_demod_wave.hip a-7: appm from the change of
// norm_rx_timing when it is below 0.2, ppm = 0.9 ppm + 0.1 appm); a row whose noise power is exactly 0 is a frame the demodulator
// skipped (non-finite input): it changed neither value. One workgroup: 256 frames at a time are fetched and prepared by all threads
// (the per-frame term 0.1 appm needs only the timing column), one thread runs the two-operation recursion over them from LDS.
__global__ __launch_bounds__(256) void ppm_kernel(float *stats, int64_t nframes, const StreamScalars *at_entry, StreamScalars *final_sc, int Nsym)
{
    __shared__ float s_nrt[257], s_term[256], s_ppm[256];
    __shared__ unsigned char s_good[257], s_upd[256];
    __shared__ float s_carry_ppm, s_carry_prev;
    const int t = threadIdx.x;
    if (t == 0) { s_carry_ppm = at_entry->ppm; s_carry_prev = at_entry->norm_rx_timing; }
    __syncthreads();
    for (int64_t f0 = 0; f0 < nframes; f0 += 256) {
        const int n = (int)(nframes - f0 < 256 ? nframes - f0 : 256);
        if (t < n) {
            const float *row = stats + (size_t)(f0 + t) * PIRIP_STATS_PER_FRAME;
            s_nrt[t + 1] = row[4];
            s_good[t + 1] = row[9] != 0.0f;
        }
        if (t == 0) { s_nrt[0] = s_carry_prev; s_good[0] = 1; }
        __syncthreads();
        // the timing value before frame t: the nearest earlier good frame's (almost always the frame before)
        if (t < n) {
            int j = t;                                    // index into s_nrt of the previous good value
            while (j > 0 && !s_good[j]) j--;
            const float d_norm = s_nrt[t + 1] - s_nrt[j];
            const bool upd = s_good[t + 1] && fabsf(d_norm) < 0.2f;
            s_upd[t] = upd;
            s_term[t] = 0.1f * ((1e6f * d_norm) / (float)Nsym);
        }
        __syncthreads();
        if (t == 0) {
            float ppm = s_carry_ppm;
            for (int i = 0; i < n; i++) {
                if (s_upd[i]) ppm = (0.9f * ppm) + s_term[i];
                s_ppm[i] = ppm;
            }
            s_carry_ppm = ppm;
            int j = n;
            while (j > 0 && !s_good[j]) j--;
            s_carry_prev = s_nrt[j];
        }
        __syncthreads();
        if (t < n) stats[(size_t)(f0 + t) * PIRIP_STATS_PER_FRAME + 7] = s_ppm[t];
        __syncthreads();
    }
    if (t == 0) final_sc->ppm = s_carry_ppm;
}

void release(CaptureWork *w)
{
    if (!w) return;
    void *ptrs[] = {w->d_Sf, w->d_theta, w->d_hist, w->d_scal, w->d_segA, w->d_segA2, w->d_segB, w->d_consA, w->d_consB, w->d_posB, w->d_nfA, w->d_nfB,
                    w->d_ok, w->d_mode, w->d_stats, w->d_warm_stats, w->d_scal0, w->d_in, w->d_bits, w->d_filt, w->d_ostats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete w;
}

int ensure_work(pirip_hip_demod *h)
{
    if (h->capture && h->capture->slots == h->nstreams) return PIRIP_OK;
    release(h->capture);
    h->capture = nullptr;
    CaptureWork *w = new (std::nothrow) CaptureWork();
    if (!w) return PIRIP_ERR_NOMEM;
    const FskDims &d = h->plan.d;
    const size_t ns = (size_t)h->nstreams;
    w->slots = h->nstreams;
    bool ok = true;
    ok &= hipMalloc((void **)&w->d_Sf, sizeof(float) * ns * d.Ndft) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_theta, sizeof(uint32_t) * ns * kMaxTones) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_hist, sizeof(float2) * ns * d.M * d.hist_len) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_scal, sizeof(StreamScalars) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_segA, sizeof(SegDesc) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_segB, sizeof(SegDesc) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_consA, sizeof(int64_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_consB, sizeof(int64_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_posB, sizeof(int64_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_nfA, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_nfB, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_ok, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_segA2, sizeof(SegDesc) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_mode, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_scal0, sizeof(StreamScalars)) == hipSuccess;
    if (!ok) { release(w); return PIRIP_ERR_NOMEM; }
    h->capture = w;
    return PIRIP_OK;
}

int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

}  // namespace

namespace pirip {
void capture_release(pirip_hip_demod *h) { if (h) { release(h->capture); h->capture = nullptr; } }
}

extern "C" int pirip_hip_demod_capture(pirip_hip_demod *h, const void *d_in, int64_t nsamp, uint8_t *d_bits, float *d_rx_filt, float *d_stats,
                                       int64_t max_frames, int64_t *nframes_out, int64_t *consumed_out, pirip_capture_report *rep,
                                       void *hip_stream)
{
    if (!h || !d_in || nsamp < 0 || max_frames <= 0) return PIRIP_ERR_BAD_ARG;
    if (!demod_bind(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const FskDims &d = h->plan.d;
    pirip_capture_report r;
    memset(&r, 0, sizeof(r));
    if (nframes_out) *nframes_out = 0;
    if (consumed_out) *consumed_out = 0;

    int rc = ensure_work(h);
    if (rc != PIRIP_OK) return rc;
    CaptureWork *w = h->capture;

    DemodArgs a;
    demod_fill_args(h, &a);
    const int Ndft = d.Ndft, hist_elems = d.M * d.hist_len, N = d.N;
    const DemodState state = a.s;
    const DemodState snap{w->d_Sf, w->d_theta, w->d_hist, w->d_scal};

    // the stream's state at entry: nin decides how many frames fit; ppm / timing seed the ppm recomputation
    StreamScalars sc0;
    CAPCHK(hipMemcpyAsync(&sc0, h->d_scal, sizeof(sc0), hipMemcpyDeviceToHost, st));
    CAPCHK(hipMemcpyAsync(w->d_scal0, h->d_scal, sizeof(sc0), hipMemcpyDeviceToDevice, st));
    CAPCHK(hipStreamSynchronize(st));
    int64_t est_frames = nsamp >= sc0.nin ? 1 + (nsamp - sc0.nin) / N : 0;
    est_frames = std::min(est_frames, max_frames);

    // segment length: a multiple of Ndft / gcd(N, Ndft) frames, so that an oscillator on an FFT bin is back at the same phase at
    // every segment start while nin = N (the first pass's phase guess is then exact for the peak estimator)
    const int G = Ndft / gcd_int(N, Ndft);
    // ... and long enough for the warm-up (= one segment) to forget its cold start: Sf's one-pole average loses a factor 0.9 per FFT, so
    // ~160 FFTs bring two histories within an ulp of each other and ~200 more let the roundings merge them for good (a bin still apart after
    // that costs one more pass, not a wrong result)
    const int nfft = std::max(1, (N - d.Ts / 4) / (Ndft / 2) - 1);
    const char *ef = getenv("PIRIP_CAPTURE_SEG_FRAMES");
    int64_t Fmin = std::max<int64_t>(ef ? atoi(ef) : 128, (400 + nfft - 1) / nfft);
    int64_t F = std::max<int64_t>(Fmin, (est_frames + h->nstreams - 1) / h->nstreams);
    F = (F + G - 1) / G * G;
    int S = (int)std::min<int64_t>(h->nstreams, (est_frames + F - 1) / F);
    const bool parallel = h->kernel == PIRIP_KERNEL_WAVE && S >= 3 && !getenv("PIRIP_CAPTURE_SEQUENTIAL") &&
                          nsamp <= demod_wave_max_samples(d);
    r.segment_frames = parallel ? (int)F : 0;
    r.segments = parallel ? S : 1;

    // per-frame statistics are always collected (the ppm column is recomputed from them): the caller's array or our own
    float *stats = d_stats;
    if (!stats) {
        // (no more rows than the samples can make frames of, whatever room the caller claims to have)
        const size_t rows = (size_t)std::min<int64_t>(max_frames, nsamp / (N - d.Ts / 4) + 2);
        if (w->stats_rows < rows) {
            if (w->d_stats) (void)hipFree(w->d_stats);
            w->d_stats = nullptr; w->stats_rows = 0;
            CAPCHK(hipMalloc((void **)&w->d_stats, sizeof(float) * rows * PIRIP_STATS_PER_FRAME + 16));
            w->stats_rows = rows;
        }
        stats = w->d_stats;
    }

    if (parallel && w->warm_rows < (size_t)S * (size_t)F) {
        if (w->d_warm_stats) (void)hipFree(w->d_warm_stats);
        w->d_warm_stats = nullptr; w->warm_rows = 0;
        CAPCHK(hipMalloc((void **)&w->d_warm_stats, sizeof(float) * (size_t)S * (size_t)F * PIRIP_STATS_PER_FRAME + 16));
        w->warm_rows = (size_t)S * (size_t)F;
    }

    if (!parallel) {
        // the sequential read loop on stream slot 0 (any kernel, any length)
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits, 0, d_rx_filt, 0, stats, 0, w->d_nfB, w->d_consB, max_frames,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, nullptr};
        hipError_t e;
        if (h->kernel == PIRIP_KERNEL_WAVE) {
            if (nsamp > demod_wave_max_samples(d)) return PIRIP_ERR_UNSUPPORTED;
            e = launch_demod_wave(a, 1, st);
        } else e = launch_demod_general(a, 1, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        int32_t nf = 0; int64_t cons = 0;
        CAPCHK(hipMemcpyAsync(&nf, w->d_nfB, sizeof(nf), hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(&cons, w->d_consB, sizeof(cons), hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        r.passes = 1; r.frames_demodulated = nf;
        if (nframes_out) *nframes_out = nf;
        if (consumed_out) *consumed_out = cons;
        if (rep) *rep = r;
        return PIRIP_OK;
    }

    // ---- frame-parallel ------------------------------------------------------------------------------------------------------
    std::vector<SegDesc> segA((size_t)h->nstreams), segA2((size_t)h->nstreams), segB((size_t)h->nstreams);
    std::vector<int64_t> pos((size_t)S, 0), len((size_t)S, 0);          // latest run of each segment: first sample, samples consumed
    std::vector<int32_t> nfr((size_t)S, 0), ok((size_t)S, 0);
    const bool debug = getenv("PIRIP_CAPTURE_DEBUG") != nullptr;
    auto seg_budget = [&](int s) -> int64_t { return s == S - 1 ? max_frames - (int64_t)s * F : F; };
    // the first frames of a warm-up run with nin pinned to N: a cold start's first tone estimates (one frame of FFTs, no integrator
    // memory) can put the timing estimate anywhere, and a timing step taken on that moves the warm-up onto another frame grid for good
    const char *ep = getenv("PIRIP_CAPTURE_PIN_FRAMES");
    const int K = std::max(0, std::min<int>((int)F / 2, ep ? atoi(ep) : 4));

    // Pilot: where does the timing loop put the frame grid? A few frames from the stream's true state on a scratch slot (state and
    // outputs untouched) -- the warm-ups then start on that grid instead of the nominal one. It matters: a cold start that finds the
    // symbol timing near the +-1/4-symbol threshold goes round either way, and every warm-up that ends a whole symbol away from its
    // neighbours is a segment to repair.
    int64_t grid0 = 0;
    {
        const int Kp = 8;
        for (int s = 0; s < h->nstreams; s++) segA[s] = SegDesc{0, 0, -1, 0};
        segA[1] = SegDesc{0, 0, Kp, 0};
        CAPCHK(hipMemcpyAsync(w->d_segA, segA.data(), sizeof(SegDesc) * segA.size(), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, 1, 0, Ndft, hist_elems);
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, nullptr, 0, w->d_nfA, w->d_consA, Kp,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA};
        hipError_t e = launch_demod_wave(a, 2, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        int64_t c = 0; int32_t nf = 0;
        CAPCHK(hipMemcpyAsync(&c, w->d_consA + 1, sizeof(c), hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(&nf, w->d_nfA + 1, sizeof(nf), hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        if (nf == Kp) grid0 = c - (int64_t)Kp * N;
        r.frames_demodulated += nf;
    }

    int v = 0;                 // segments < v are final (verified chain from the stream's true state)
    int64_t total_frames = 0, total_consumed = 0;
    int final_slot = 0;
    std::vector<int32_t> mode((size_t)h->nstreams);          // this pass: 0 sits out, 1 warms up from a cold state, 2 continues from its predecessor's end state
    for (;;) {
        r.passes++;
        // Pass 1: segment 0 from the stream's true state, every other segment from a cold warm-up over its predecessor's samples on the
        // pilot's grid. Later passes: segment v continues from the verified end state of segment v-1, and everything after it is
        // speculated again, first samples from the prefix sums of the lengths the segments measured last time. (Measured against
        // repairing locally -- re-running only the segments whose start failed, from their predecessor's end state, and keeping the rest:
        // what breaks a link is almost always the demodulator's own timing loop slipping a WHOLE symbol inside a segment, which a cold
        // warm-up over the same samples resolves the other way half the time; after that the chain is a symbol away from everything
        // speculated downstream and never re-joins it. Local repair then costs a pass to find that out: 21 passes instead of 9 at 7 dB.)
        int nrun = 0;
        bool any_warm = false, any_cont = false;
        for (int s = 0; s < h->nstreams; s++) { segA[s] = SegDesc{0, 0, -1, 0}; segA2[s] = SegDesc{0, 0, -1, 0}; segB[s] = SegDesc{0, 0, -1, 0}; mode[s] = 0; }
        auto budget32 = [&](int s) { return (int32_t)std::min<int64_t>(seg_budget(s), 0x7fffffff); };
        auto warm = [&](int s, int64_t start) {      // cold warm-up of segment s over its predecessor's samples, first sample `start`
            const int64_t p0 = std::max<int64_t>(0, std::min(start, nsamp));          // beyond the data: nothing to do, the segment stays empty
            segA[s] = SegDesc{p0, (int64_t)s * F, K, 0};
            segA2[s] = SegDesc{std::min(p0 + (int64_t)K * N, nsamp), (int64_t)s * F + K, (int32_t)F - K, 0};
            segB[s] = SegDesc{0, (int64_t)s * F, budget32(s), 0};
            mode[s] = 1; any_warm = true; nrun++;
        };
        auto cont = [&](int s) {
            segB[s] = SegDesc{0, (int64_t)s * F, budget32(s), 0};                       // first sample: filled in on the device
            mode[s] = 2; any_cont = true; nrun++;
        };
        if (r.passes == 1) {
            segB[0] = SegDesc{0, 0, budget32(0), 0};
            nrun++;
            for (int s = 1; s < S; s++) warm(s, (int64_t)(s - 1) * F * N + (s > 1 ? grid0 : 0));
        } else {
            cont(v);
            int64_t g = pos[v - 1] + len[v - 1];              // first sample of segment v (exact)
            for (int s = v + 1; s < S; s++) { warm(s, g); g += len[s - 1]; }
        }
        CAPCHK(hipMemcpyAsync(w->d_segA, segA.data(), sizeof(SegDesc) * segA.size(), hipMemcpyHostToDevice, st));
        CAPCHK(hipMemcpyAsync(w->d_segA2, segA2.data(), sizeof(SegDesc) * segA2.size(), hipMemcpyHostToDevice, st));
        CAPCHK(hipMemcpyAsync(w->d_segB, segB.data(), sizeof(SegDesc) * segB.size(), hipMemcpyHostToDevice, st));
        CAPCHK(hipMemcpyAsync(w->d_mode, mode.data(), sizeof(int32_t) * mode.size(), hipMemcpyHostToDevice, st));
        hipError_t e = hipSuccess;
        if (any_warm) {
            hipLaunchKernelGGL(cold_kernel, dim3(S), dim3(kThreads), 0, st, state, (const SegDesc *)w->d_segA, Ndft, hist_elems, N);
            // warm-up launches (statistics rows to a scratch array, never read)
            if (K > 0) {
                DemodArgs ap = a;
                ap.d.burst_mode = 1;                       // fsk_enable_burst_mode(): nin stays N
                ap.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, w->d_warm_stats, 0, w->d_nfA, w->d_consA, F,
                                SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA};
                e = launch_demod_wave(ap, S, st);
                if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
            }
            a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, w->d_warm_stats, 0, w->d_nfA, w->d_consA, F,
                           SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA2};
            e = launch_demod_wave(a, S, st);
            if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
            hipLaunchKernelGGL(after_warmup_kernel, dim3(S), dim3(kThreads), 0, st, state, snap, (const SegDesc *)w->d_segA2, w->d_segB,
                               (const int64_t *)w->d_consA, w->d_posB, Ndft, hist_elems, r.passes == 1 ? 0 : -1, (int64_t)0);
        }
        if (any_cont)
            hipLaunchKernelGGL(continue_kernel, dim3(S), dim3(kThreads), 0, st, state, snap, (const int32_t *)w->d_mode, w->d_segB,
                               (const int64_t *)w->d_consB, w->d_posB, Ndft, hist_elems);
        // the segments' own frames: outputs to their rows of the capture's arrays
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits, 0, d_rx_filt, 0, stats, 0, w->d_nfB, w->d_consB, F,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segB};
        e = launch_demod_wave(a, S, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        if (S - 1 - v > 0)
            hipLaunchKernelGGL(verify_kernel, dim3(S - 1 - v), dim3(kThreads), 0, st, state, snap, (const int64_t *)w->d_posB,
                               (const int64_t *)w->d_consB, w->d_ok, v + 1, Ndft, hist_elems, d.M);
        CAPCHK(hipGetLastError());
        // what the host needs for the next decision
        std::vector<int64_t> h_pos((size_t)S), h_len((size_t)S);
        std::vector<int32_t> h_nf((size_t)S), h_ok((size_t)S, 0);
        CAPCHK(hipMemcpyAsync(h_pos.data(), w->d_posB, sizeof(int64_t) * S, hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(h_len.data(), w->d_consB, sizeof(int64_t) * S, hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(h_nf.data(), w->d_nfB, sizeof(int32_t) * S, hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(h_ok.data(), w->d_ok, sizeof(int32_t) * S, hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        for (int s = 0; s < S; s++) {
            pos[s] = h_pos[s]; len[s] = h_len[s]; nfr[s] = h_nf[s];
            if ((r.passes == 1 && s == 0) || mode[s]) r.frames_demodulated += h_nf[s] + (mode[s] == 1 ? F : 0);
            if (s > v) ok[s] = h_ok[s] == 0;
        }
        if (debug) {
            int nbad = 0, first = -1, why = 0;
            for (int s = v + 1; s < S; s++) if (!ok[s]) { if (first < 0) { first = s; why = h_ok[s]; } nbad++; }
            int hist[6] = {0, 0, 0, 0, 0, 0};
            for (int s = v + 1; s < S; s++) for (int b = 0; b < 6; b++) if (h_ok[s] & (1 << b)) hist[b]++;
            fprintf(stderr, "capture pass %d (%s): v = %d, ran %d of %d segments, %d starts unverified (first: segment %d, mask %d); differing: Sf %d tail %d phase %d nin %d timing %d position %d\n",
                    r.passes, r.passes == 1 ? "first" : "again from the verified chain", v, nrun, S, nbad, first, why, hist[0], hist[1], hist[2],
                    hist[3], hist[4], hist[5]);
            if (first > 0) {
                fprintf(stderr, "   unverified starts (segment: its first sample minus its predecessor's end):");
                int shown = 0;
                for (int s2 = v + 1; s2 < S && shown < 16; s2++)
                    if (!ok[s2]) { fprintf(stderr, " %d: %+lld (%d)", s2, (long long)(pos[s2] - (pos[s2 - 1] + len[s2 - 1])), h_ok[s2]); shown++; }
                fprintf(stderr, "\n");
            }
        }
        // advance over everything that is now verified
        int nv = v + 1;
        while (nv < S && nfr[nv - 1] == F && ok[nv]) nv++;
        if (r.passes > 1) r.segments_rerun += nrun;
        if (nv == S || nfr[nv - 1] < F) {                  // the last segment, or the one the samples (or the output rows) ran out in
            final_slot = nv - 1;
            total_frames = (int64_t)(nv - 1) * F + nfr[nv - 1];
            total_consumed = pos[nv - 1] + len[nv - 1];
            break;
        }
        v = nv;
    }
    // the capture's end state becomes the stream's (slot 0), with ppm recomputed in frame order
    if (final_slot != 0) hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, 0, final_slot, Ndft, hist_elems);
    hipLaunchKernelGGL(ppm_kernel, dim3(1), dim3(256), 0, st, stats, total_frames, (const StreamScalars *)w->d_scal0, h->d_scal, d.Nsym);
    CAPCHK(hipGetLastError());
    CAPCHK(hipStreamSynchronize(st));
    if (nframes_out) *nframes_out = total_frames;
    if (consumed_out) *consumed_out = total_consumed;
    if (rep) *rep = r;
    return PIRIP_OK;
}

// Host-buffer form (what the fsk_demod tool calls for a file): upload, pirip_hip_demod_capture, download.
extern "C" int pirip_hip_demod_capture_host(pirip_hip_demod *h, const void *in, int64_t nsamp, uint8_t *bits, float *rx_filt, float *stats,
                                            int64_t max_frames, int64_t *nframes_out, int64_t *consumed_out, pirip_capture_report *rep)
{
    if (!h || (!in && nsamp > 0) || nsamp < 0 || max_frames <= 0 || !bits) return PIRIP_ERR_BAD_ARG;
    if (!demod_bind(h)) return PIRIP_ERR_NO_DEVICE;
    int rc = ensure_work(h);
    if (rc != PIRIP_OK) return rc;
    CaptureWork *w = h->capture;
    const FskDims &d = h->plan.d;
    const size_t bps = d.in_format == PIRIP_IN_CF32 ? 8 : d.in_format == PIRIP_IN_CS16 ? 4 : 2;
    const size_t in_bytes = (size_t)nsamp * bps;
    if (in_bytes > w->in_bytes || !w->d_in) {
        if (w->d_in) (void)hipFree(w->d_in);
        w->d_in = nullptr; w->in_bytes = 0;
        CAPCHK(hipMalloc(&w->d_in, in_bytes + 64));
        w->in_bytes = in_bytes;
    }
    const size_t fb = d.pack_bits ? (size_t)(d.Nbits + 7) / 8 : (size_t)d.Nbits;
    if (max_frames > w->out_frames) {
        void *olds[] = {w->d_bits, w->d_filt, w->d_ostats};
        for (void *p : olds) if (p) (void)hipFree(p);
        w->d_bits = nullptr; w->d_filt = nullptr; w->d_ostats = nullptr; w->out_frames = 0;
        CAPCHK(hipMalloc((void **)&w->d_bits, (size_t)max_frames * fb + 16));
        CAPCHK(hipMalloc((void **)&w->d_filt, sizeof(float) * (size_t)max_frames * d.M * d.Nsym + 16));
        CAPCHK(hipMalloc((void **)&w->d_ostats, sizeof(float) * (size_t)max_frames * PIRIP_STATS_PER_FRAME + 16));
        w->out_frames = max_frames;
    }
    if (in_bytes) CAPCHK(hipMemcpy(w->d_in, in, in_bytes, hipMemcpyHostToDevice));
    int64_t nf = 0, cons = 0;
    rc = pirip_hip_demod_capture(h, w->d_in, nsamp, w->d_bits, rx_filt ? w->d_filt : nullptr, w->d_ostats, max_frames, &nf, &cons, rep, nullptr);
    if (rc != PIRIP_OK) return rc;
    if (nf) CAPCHK(hipMemcpy(bits, w->d_bits, (size_t)nf * fb, hipMemcpyDeviceToHost));
    if (rx_filt && nf) CAPCHK(hipMemcpy(rx_filt, w->d_filt, sizeof(float) * (size_t)nf * d.M * d.Nsym, hipMemcpyDeviceToHost));
    if (stats && nf) CAPCHK(hipMemcpy(stats, w->d_ostats, sizeof(float) * (size_t)nf * PIRIP_STATS_PER_FRAME, hipMemcpyDeviceToHost));
    StreamScalars sc;
    CAPCHK(hipMemcpy(&sc, h->d_scal, sizeof(sc), hipMemcpyDeviceToHost));
    h->nin0 = sc.nin;
    if (nframes_out) *nframes_out = nf;
    if (consumed_out) *consumed_out = cons;
    return PIRIP_OK;
}
