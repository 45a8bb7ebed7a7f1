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
// Segment 0 starts from the handle's true state. What breaks the chain, measured: a sample-clock offset (every quarter-symbol step of
// the timing loop moves the frame grid of everything after it: the guessed first samples have to be refreshed from the lengths the
// segments measured), and -- with noise -- the timing loop going once ROUND, a net step of a whole symbol, after which the chain sees
// exactly the timing everything speculated downstream sees, one symbol apart, and never meets it again. So after the first pass every
// segment is speculated on several frame grids a whole symbol apart (replicas; more of them the further from the verified chain's end,
// as a random walk needs, until the stream slots are used up), and the chain continues through whichever replica starts at the sample
// and in the state it ended in. Each pass verifies at least one more segment, so the worst case is the sequential loop's cost (x2 for
// the warm-ups); the usual case is one pass, or a few. Every pass writes to the slots' own output rows; the verified segments' rows are
// copied to the caller's arrays.
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
    int32_t *d_nfA = nullptr, *d_nfB = nullptr, *d_ok = nullptr, *d_mode = nullptr, *d_src = nullptr, *d_from = nullptr;
    // the replicas' own output rows (passes after the first: several candidates per segment, the verified one is copied out)
    uint8_t *r_bits = nullptr; float *r_filt = nullptr, *r_stats = nullptr; size_t r_rows = 0; bool r_has_filt = false;
    float *d_stats = nullptr; size_t stats_rows = 0;
    float *d_warm_stats = nullptr; size_t warm_rows = 0;   // statistics rows of the warm-up frames (never read: they make every warm-up frame
                                                           // an observable one, so that snr_est's average runs through them)
    StreamScalars *d_scal0 = nullptr;            // the stream's scalars at entry (ppm / timing the recomputation starts from)
    // staging of the host-buffer form
    void *d_in = nullptr; size_t in_bytes = 0;
    uint8_t *d_bits = nullptr; float *d_filt = nullptr; float *d_ostats = nullptr; int64_t out_frames = 0;
};

namespace {

// Host-to-device copy whose source is a stack variable or a vector that the next pass rewrites: complete before returning, whatever
// the runtime does with pageable sources
#define CAP_H2D(dst, src, bytes, st) do { CAPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); CAPCHK(hipStreamSynchronize(st)); } while (0)
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
                                    int Ndft, int hist_elems)
{
    const int s = blockIdx.x;
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

// the segment at the head of the unverified part: its start state is the (verified) end state of its predecessor (slot src[q]), its first
// sample the predecessor's last + 1. The predecessor does not run in this pass.
__global__ void continue_kernel(DemodState st, DemodState snap, const int32_t *mode, const int32_t *src, SegDesc *segB, const int64_t *consB,
                                int64_t *posB, int Ndft, int hist_elems)
{
    const int q = blockIdx.x;
    if (mode[q] != 2) return;
    const int p = src[q];
    for (int i = threadIdx.x; i < Ndft; i += blockDim.x) {
        const float x = st.Sf[(size_t)p * Ndft + i];
        st.Sf[(size_t)q * Ndft + i] = x; snap.Sf[(size_t)q * Ndft + i] = x;
    }
    for (int i = threadIdx.x; i < hist_elems; i += blockDim.x) {
        const float2 x = st.hist[(size_t)p * hist_elems + i];
        st.hist[(size_t)q * hist_elems + i] = x; snap.hist[(size_t)q * hist_elems + i] = x;
    }
    if (threadIdx.x < kMaxTones) {
        const uint32_t x = st.theta[(size_t)p * kMaxTones + threadIdx.x];
        st.theta[(size_t)q * kMaxTones + threadIdx.x] = x; snap.theta[(size_t)q * kMaxTones + threadIdx.x] = x;
    }
    if (threadIdx.x == 0) {
        const StreamScalars x = st.scal[p];
        st.scal[q] = x; snap.scal[q] = x;
        const int64_t at = posB[p] + consB[p];
        posB[q] = at;
        segB[q].in_off = at;
    }
}

// ok[i] = 0 when the state slot pq[2i+1] started its own frames from is, bit for bit, the state slot pq[2i] ended in, at the same sample;
// otherwise what differs: 1 Sf, 2 integrator tail, 4 oscillator phase, 8 nin, 16 timing, 32 sample position, 64 another carried scalar
__global__ void verify_kernel(DemodState st, DemodState snap, const int64_t *posB, const int64_t *consB, const int32_t *pq, int32_t *ok, int Ndft,
                              int hist_elems, int M, int sf_lo, int sf_hi)      // [sf_lo, sf_hi): the Sf bins the handle maintains (all, or the band-only estimator's)
{
    const int p = pq[2 * blockIdx.x], q = pq[2 * blockIdx.x + 1];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = posB[p] + consB[p] != posB[q] ? 32 : 0;
    __syncthreads();
    int b = 0;
    const uint32_t *a0 = (const uint32_t *)(st.Sf + (size_t)p * Ndft), *b0 = (const uint32_t *)(snap.Sf + (size_t)q * Ndft);
    for (int i = sf_lo + threadIdx.x; i < sf_hi; i += blockDim.x) b |= a0[i] != b0[i] ? 1 : 0;
    const uint32_t *a1 = (const uint32_t *)(st.hist + (size_t)p * hist_elems), *b1 = (const uint32_t *)(snap.hist + (size_t)q * hist_elems);
    for (int i = threadIdx.x; i < 2 * hist_elems; i += blockDim.x) b |= a1[i] != b1[i] ? 2 : 0;
    if (threadIdx.x < M) b |= st.theta[(size_t)p * kMaxTones + threadIdx.x] != snap.theta[(size_t)q * kMaxTones + threadIdx.x] ? 4 : 0;
    if (threadIdx.x == 0) {
        const StreamScalars x = st.scal[p], y = snap.scal[q];
        b |= x.nin != y.nin ? 8 : 0;
        b |= __float_as_uint(x.norm_rx_timing) != __float_as_uint(y.norm_rx_timing) ? 16 : 0;
        // every other carried scalar too (snr_est is a 0.5/0.5 IIR over frames, SNRest / EbNodB / v_est / f_est / rx_*_pow are what
        // pirip_hip_get_stream_state and a skipped frame's stats row show): a segment is only accepted when the state it started
        // from is the state its predecessor ended in, word for word. ppm is left out: the caller recomputes it along the chain.
        const uint32_t *xs = (const uint32_t *)&x, *ys = (const uint32_t *)&y;
        constexpr int kPpmWord = offsetof(StreamScalars, ppm) / 4;
        for (int i = 0; i < (int)(sizeof(StreamScalars) / 4); i++) if (i != kPpmWord && xs[i] != ys[i]) b |= 64;
    }
    if (b) atomicOr(&bad, b);
    __syncthreads();
    if (threadIdx.x == 0) ok[blockIdx.x] = bad;
}

// rows of the verified segments from the replicas' own output arrays to the caller's: segment first + blockIdx.x came out of slot from[s]
__global__ void gather_kernel(const int32_t *from, const int32_t *nfr, int first, int F, int frame_bytes, int filt_floats, const uint8_t *rbits,
                              uint8_t *bits, const float *rfilt, float *filt, const float *rstats, float *stats)
{
    const int s = first + blockIdx.x, q = from[s], n = nfr[q];
    const size_t src = (size_t)q * F, dst = (size_t)s * F;
    if (blockIdx.y == 0 && bits) {
        const size_t nb = (size_t)n * frame_bytes;
        for (size_t i = threadIdx.x; i < nb; i += blockDim.x) bits[dst * frame_bytes + i] = rbits[src * frame_bytes + i];
    } else if (blockIdx.y == 1 && stats) {
        const size_t nb = (size_t)n * PIRIP_STATS_PER_FRAME;
        for (size_t i = threadIdx.x; i < nb; i += blockDim.x) stats[dst * PIRIP_STATS_PER_FRAME + i] = rstats[src * PIRIP_STATS_PER_FRAME + i];
    } else if (blockIdx.y == 2 && filt) {
        const size_t nb = (size_t)n * filt_floats;
        for (size_t i = threadIdx.x; i < nb; i += blockDim.x) filt[dst * filt_floats + i] = rfilt[src * filt_floats + i];
    }
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
                    w->d_ok, w->d_mode, w->d_src, w->d_from, w->r_bits, w->r_filt, w->r_stats, w->d_stats, w->d_warm_stats, w->d_scal0, w->d_in, w->d_bits, w->d_filt, w->d_ostats};
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
    ok &= hipMalloc((void **)&w->d_ok, sizeof(int32_t) * ns * 4) == hipSuccess;     // pair list (2 ints per link), then the verdicts
    ok &= hipMalloc((void **)&w->d_src, sizeof(int32_t) * ns) == hipSuccess;
    ok &= hipMalloc((void **)&w->d_from, sizeof(int32_t) * ns) == hipSuccess;
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

static int capture_impl(pirip_hip_demod *h, const void *d_in, int64_t nsamp, uint8_t *d_bits, float *d_rx_filt, float *d_stats,
                        int64_t max_frames, int64_t *nframes_out, int64_t *consumed_out, pirip_capture_report *rep, void *hip_stream);

// Public entry. A stream still in its created state hands its FIRST frame to the exact prologue where the read loop does
// (pirip_capi.hip: exact0_prologue; shapes with P == Ts), so that the capture stays bit for bit what the read loop gives; the capture
// proper (capture_impl, unchanged since round 4) then starts one frame in.
extern "C" int pirip_hip_demod_capture(pirip_hip_demod *h, const void *d_in, int64_t nsamp, uint8_t *d_bits, float *d_rx_filt, float *d_stats,
                                       int64_t max_frames, int64_t *nframes_out, int64_t *consumed_out, pirip_capture_report *rep,
                                       void *hip_stream)
{
    if (!h || !d_in || nsamp < 0 || max_frames <= 0) return PIRIP_ERR_BAD_ARG;
    if (!demod_bind(h)) return PIRIP_ERR_NO_DEVICE;
    const FskDims &d = h->plan.d;
    const bool was_fresh = h->fresh;
    if (nsamp >= d.N) h->fresh = false;
    if (!(was_fresh && nsamp >= d.N && h->exact0 && demod_exact0_applicable(d) && h->kernel != PIRIP_KERNEL_BLOCK && h->kernel != PIRIP_KERNEL_EXACT))
        return capture_impl(h, d_in, nsamp, d_bits, d_rx_filt, d_stats, max_frames, nframes_out, consumed_out, rep, hip_stream);
    hipStream_t st = (hipStream_t)hip_stream;
    DemodArgs a;
    demod_fill_args(h, &a);
    a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits, 0, d_rx_filt, 0, d_stats, 0, nullptr, nullptr, 1, SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, nullptr};
    a.io.first_out = h->d_first;
    a.io.exact0_fmt = h->kernel == PIRIP_KERNEL_WAVE ? PIRIP_KERNEL_WAVE : PIRIP_KERNEL_GENERAL;
    const hipError_t e = launch_demod_exact0(a, 1, st);               // stream slot 0 is the capture's stream
    if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
    int32_t n0 = 0;
    CAPCHK(hipMemcpyAsync(&n0, h->d_first, sizeof(n0), hipMemcpyDeviceToHost, st));
    CAPCHK(hipStreamSynchronize(st));
    if (n0 <= 0) return capture_impl(h, d_in, nsamp, d_bits, d_rx_filt, d_stats, max_frames, nframes_out, consumed_out, rep, hip_stream);
    const size_t bps = d.in_format == PIRIP_IN_CF32 ? 8 : d.in_format == PIRIP_IN_CS16 ? 4 : 2;
    const size_t fb = d.pack_bits ? (size_t)(d.Nbits + 7) / 8 : (size_t)d.Nbits;
    int64_t nf = 0, cons = 0;
    pirip_capture_report r;
    memset(&r, 0, sizeof(r));
    int rc = PIRIP_OK;
    if (max_frames > 1)
        rc = capture_impl(h, (const uint8_t *)d_in + (size_t)n0 * bps, nsamp - n0, d_bits ? d_bits + fb : nullptr, d_rx_filt ? d_rx_filt + (size_t)d.M * d.Nsym : nullptr,
                          d_stats ? d_stats + PIRIP_STATS_PER_FRAME : nullptr, max_frames - 1, &nf, &cons, &r, hip_stream);
    if (rc != PIRIP_OK) return rc;
    r.frames_demodulated += 1;
    if (nframes_out) *nframes_out = nf + 1;
    if (consumed_out) *consumed_out = cons + n0;
    if (rep) *rep = r;
    return PIRIP_OK;
}

static int capture_impl(pirip_hip_demod *h, const void *d_in, int64_t nsamp, uint8_t *d_bits, float *d_rx_filt, float *d_stats,
                        int64_t max_frames, int64_t *nframes_out, int64_t *consumed_out, pirip_capture_report *rep, void *hip_stream)
{
    if (!h || !d_in || nsamp < 0 || max_frames <= 0) return PIRIP_ERR_BAD_ARG;
    if (!demod_bind(h)) return PIRIP_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    const FskDims &d = h->plan.d;
    pirip_capture_report r;
    memset(&r, 0, sizeof(r));
    if (nframes_out) *nframes_out = 0;
    if (consumed_out) *consumed_out = 0;

    const int err = ensure_work(h);
    if (err != PIRIP_OK) return err;
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
    const int nfft = std::max(1, (N - d.nin_step) / (Ndft / 2) - 1);
    const char *ef = getenv("PIRIP_CAPTURE_SEG_FRAMES");
    int64_t Fmin = std::max<int64_t>(ef ? atoi(ef) : 128, (400 + nfft - 1) / nfft);
    // (one stream slot holds the verified chain's end state between passes; the others are shared out, pass by pass, as replicas of segments)
    const int usable = h->nstreams - 1;
    int64_t F = std::max<int64_t>(Fmin, usable > 0 ? (est_frames + usable - 1) / usable : est_frames);
    F = (F + G - 1) / G * G;
    int S = (int)std::min<int64_t>(std::max(usable, 0), (est_frames + F - 1) / F);
    const bool parallel = h->kernel == PIRIP_KERNEL_WAVE && S >= 3 && !getenv("PIRIP_CAPTURE_SEQUENTIAL") &&
                          nsamp <= demod_wave_max_samples(d);
    r.segment_frames = parallel ? (int)F : 0;
    r.segments = parallel ? S : 1;

    // per-frame statistics are always collected (the ppm column is recomputed from them): the caller's array or our own
    float *stats = d_stats;
    if (!stats) {
        // (no more rows than the samples can make frames of, whatever room the caller claims to have)
        const size_t rows = (size_t)std::min<int64_t>(max_frames, nsamp / (N - d.nin_step) + 2);
        if (w->stats_rows < rows) {
            if (w->d_stats) (void)hipFree(w->d_stats);
            w->d_stats = nullptr; w->stats_rows = 0;
            CAPCHK(hipMalloc((void **)&w->d_stats, sizeof(float) * rows * PIRIP_STATS_PER_FRAME + 16));
            w->stats_rows = rows;
        }
        stats = w->d_stats;
    }

    const size_t rrows = (size_t)h->nstreams * (size_t)F;                  // one row block of F frames per stream slot
    if (parallel && w->warm_rows < rrows) {
        if (w->d_warm_stats) (void)hipFree(w->d_warm_stats);
        w->d_warm_stats = nullptr; w->warm_rows = 0;
        CAPCHK(hipMalloc((void **)&w->d_warm_stats, sizeof(float) * rrows * PIRIP_STATS_PER_FRAME + 16));
        w->warm_rows = rrows;
    }

    if (!parallel) {
        // the sequential read loop on stream slot 0 (any kernel, any length)
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits, 0, d_rx_filt, 0, stats, 0, w->d_nfB, w->d_consB, max_frames,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, nullptr};
        a.io.eye = (h->kernel == PIRIP_KERNEL_GENERAL || h->kernel == PIRIP_KERNEL_EXACT) ? h->d_eye : nullptr;      // as pirip_hip_demod_batch: the latest frame's traces
        hipError_t e;
        if (h->kernel == PIRIP_KERNEL_WAVE) {
            if (nsamp > demod_wave_max_samples(d)) return PIRIP_ERR_UNSUPPORTED;
            e = launch_demod_wave(a, 1, st);
        } else e = launch_demod_kind(h->kernel, a, 1, st);
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
    const int KEEP = h->nstreams - 1, Ts = d.Ts;          // the slot that holds the verified chain's end state between passes
    const size_t fb = d.pack_bits ? (size_t)(d.Nbits + 7) / 8 : (size_t)d.Nbits;
    const int filt_floats = d.M * d.Nsym;
    std::vector<SegDesc> segA((size_t)h->nstreams), segA2((size_t)h->nstreams), segB((size_t)h->nstreams);
    std::vector<int32_t> mode((size_t)h->nstreams), src((size_t)h->nstreams, KEEP), from((size_t)h->nstreams, 0);
    std::vector<int32_t> off_of((size_t)h->nstreams), first_slot((size_t)S + 1), nslot_of((size_t)S + 1);
    std::vector<int64_t> len_est((size_t)S, (int64_t)F * N);            // samples each segment consumes: measured where it has run, nominal before
    const bool debug = getenv("PIRIP_CAPTURE_DEBUG") != nullptr;
    // the first frames of a warm-up run with nin pinned to N: a cold start's first tone estimates (one frame of FFTs, no integrator
    // memory) can put the timing estimate anywhere, and a timing step taken on that moves the warm-up onto another frame grid for good
    const char *ep = getenv("PIRIP_CAPTURE_PIN_FRAMES");
    const int K = std::max(0, std::min<int>((int)F / 2, ep ? atoi(ep) : 4));
    const char *er = getenv("PIRIP_CAPTURE_REPLICAS");
    const int Hcap = std::max(0, ((er ? atoi(er) : 7) - 1) / 2);       // at most 2 Hcap + 1 replicas of a segment (7: measured best of 1..15)
    auto skip_all = [&]() {
        for (int q = 0; q < h->nstreams; q++) { segA[q] = SegDesc{0, 0, -1, 0}; segA2[q] = SegDesc{0, 0, -1, 0}; segB[q] = SegDesc{0, 0, -1, 0}; mode[q] = 0; }
    };
    // output rows of the slots (every pass writes there; the rows of the verified segments are copied to the caller's arrays)
    if (w->r_rows < rrows || (d_rx_filt && !w->r_has_filt)) {
        void *olds[] = {w->r_bits, w->r_filt, w->r_stats};
        for (void *o : olds) if (o) (void)hipFree(o);
        w->r_bits = nullptr; w->r_filt = nullptr; w->r_stats = nullptr; w->r_rows = 0; w->r_has_filt = false;
        CAPCHK(hipMalloc((void **)&w->r_bits, rrows * fb + 16));
        CAPCHK(hipMalloc((void **)&w->r_stats, sizeof(float) * rrows * PIRIP_STATS_PER_FRAME + 16));
        if (d_rx_filt) { CAPCHK(hipMalloc((void **)&w->r_filt, sizeof(float) * rrows * filt_floats + 16)); w->r_has_filt = true; }
        w->r_rows = rrows;
    }

    // the stream's true state is the chain's first end state (of "segment -1", ending at sample 0)
    hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, KEEP, 0, Ndft, hist_elems);

    // Pilot: where does the timing loop put the frame grid? A few frames from the stream's true state on a scratch slot (state and
    // outputs untouched) -- the warm-ups then start on that grid instead of the nominal one. It matters: a cold start that finds the
    // symbol timing near the +-1/4-symbol threshold goes round either way, and every warm-up that ends a whole symbol away from its
    // neighbours is a segment to repair.
    int64_t grid0 = 0;
    {
        const int Kp = 8;
        skip_all();
        segA[0] = SegDesc{0, 0, Kp, 0};
        CAP_H2D(w->d_segA, segA.data(), sizeof(SegDesc) * segA.size(), st);
        hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, 0, KEEP, Ndft, hist_elems);
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, nullptr, 0, w->d_nfA, w->d_consA, Kp,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA};
        hipError_t e = launch_demod_wave(a, 1, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        int64_t c = 0; int32_t nf = 0;
        CAPCHK(hipMemcpyAsync(&c, w->d_consA, sizeof(c), hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(&nf, w->d_nfA, sizeof(nf), hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        if (nf == Kp) grid0 = c - (int64_t)Kp * N;
        r.frames_demodulated += nf;
    }

    int v = 0;                         // segments < v are final (verified chain from the stream's true state)
    int64_t chain_pos = 0;             // first sample of segment v = where the verified chain ends
    int64_t total_frames = 0, total_consumed = 0;
    int symbol_moves = 0;              // whole symbols the verified chain has moved against the guesses so far
    bool more_after_last = false;      // the last segment stopped at the end of its row block, not at the end of the samples
    for (;;) {
        r.passes++;
        // Every pass: the head segment v continues from the verified chain's end state; every later segment is speculated from a cold
        // warm-up over its predecessor's samples, first samples from the prefix sums of the segments' (measured, else nominal) lengths
        // -- in pass 1 once, on the pilot's grid. What breaks the chain, measured: a sample-clock offset (every quarter-symbol step of
        // the timing loop moves the grid of everything after it -- a warm-up a quarter or half a symbol off finds its way back, the
        // guesses just have to be refreshed), and the timing loop going once ROUND: a net step of a whole symbol, after which the chain
        // sees exactly the timing everything speculated downstream sees, one symbol apart, and never meets it again. So from pass 2 on a
        // segment is speculated on several frame grids a whole symbol apart (replicas), as many as the chain's random walk may need at
        // that distance from the head -- 1 + 2 ceil(2 sqrt(distance x moves per segment so far)) -- until the stream slots are used up;
        // the chain continues through whichever replica starts at the sample and in the state it ended in.
        const double lambda = r.passes == 1 ? 0.0 : (symbol_moves + 1.0) / std::max(1, v);
        skip_all();
        int nslots = 0;
        std::vector<int64_t> gs((size_t)S + 1);              // guessed first sample of every segment from v on
        gs[v] = chain_pos;
        for (int s2 = v; s2 < S; s2++) gs[s2 + 1] = gs[s2] + (r.passes == 1 && s2 == 0 ? grid0 + len_est[0] : len_est[s2]);
        auto budget = [&](int s2) { return (int32_t)std::min<int64_t>(std::min<int64_t>(F, max_frames - (int64_t)s2 * F), 0x7fffffff); };
        // head
        {
            const int q = nslots++;
            off_of[q] = 0; first_slot[v] = q; nslot_of[v] = 1;
            segB[q] = SegDesc{0, (int64_t)q * F, budget(v), 0};
            mode[q] = 2; src[q] = KEEP;
        }
        int s_hi = v;
        for (int s2 = v + 1; s2 < S; s2++) {
            const int H = std::min<int>(Hcap, (int)std::ceil(2.0 * std::sqrt((double)(s2 - v) * lambda)));
            if (nslots + 2 * H + 1 > KEEP) break;
            first_slot[s2] = nslots; nslot_of[s2] = 2 * H + 1;
            for (int o = -H; o <= H; o++) {
                const int q = nslots++;
                off_of[q] = o;
                // cold warm-up over the predecessor's samples; beyond the data: nothing to do, the segment stays empty
                const int64_t p0 = std::max<int64_t>(0, std::min(gs[s2 - 1] + (int64_t)o * Ts, nsamp));
                segA[q] = SegDesc{p0, (int64_t)q * F, K, 0};
                segA2[q] = SegDesc{std::min(p0 + (int64_t)K * N, nsamp), (int64_t)q * F + K, (int32_t)F - K, 0};
                segB[q] = SegDesc{0, (int64_t)q * F, budget(s2), 0};
                mode[q] = 1;
            }
            s_hi = s2;
        }
        CAP_H2D(w->d_segA, segA.data(), sizeof(SegDesc) * segA.size(), st);
        CAP_H2D(w->d_segA2, segA2.data(), sizeof(SegDesc) * segA2.size(), st);
        CAP_H2D(w->d_segB, segB.data(), sizeof(SegDesc) * segB.size(), st);
        CAP_H2D(w->d_mode, mode.data(), sizeof(int32_t) * mode.size(), st);
        CAP_H2D(w->d_src, src.data(), sizeof(int32_t) * src.size(), st);
        {   // where the chain ends, for the head's first sample
            CAP_H2D(w->d_posB + KEEP, &chain_pos, sizeof(int64_t), st);
            CAPCHK(hipMemsetAsync(w->d_consB + KEEP, 0, sizeof(int64_t), st));
        }
        hipError_t e = hipSuccess;
        if (nslots > 1) {
            hipLaunchKernelGGL(cold_kernel, dim3(nslots), dim3(kThreads), 0, st, state, (const SegDesc *)w->d_segA, Ndft, hist_elems, N);
            // warm-up launches (statistics rows to a scratch array, never read)
            if (K > 0) {
                DemodArgs ap = a;
                ap.d.burst_mode = 1;                       // fsk_enable_burst_mode(): nin stays N
                ap.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, w->d_warm_stats, 0, w->d_nfA, w->d_consA, F,
                                SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA};
                e = launch_demod_wave(ap, nslots, st);
                if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
            }
            a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, nullptr, 0, nullptr, 0, w->d_warm_stats, 0, w->d_nfA, w->d_consA, F,
                           SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segA2};
            e = launch_demod_wave(a, nslots, st);
            if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
            hipLaunchKernelGGL(after_warmup_kernel, dim3(nslots), dim3(kThreads), 0, st, state, snap, (const SegDesc *)w->d_segA2, w->d_segB,
                               (const int64_t *)w->d_consA, w->d_posB, Ndft, hist_elems);
        }
        hipLaunchKernelGGL(continue_kernel, dim3(1), dim3(kThreads), 0, st, state, snap, (const int32_t *)w->d_mode, (const int32_t *)w->d_src,
                           w->d_segB, (const int64_t *)w->d_consB, w->d_posB, Ndft, hist_elems);
        // the segments' own frames, to the slots' rows
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits ? w->r_bits : nullptr, 0, d_rx_filt ? w->r_filt : nullptr, 0, w->r_stats, 0,
                       w->d_nfB, w->d_consB, F, SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segB};
        e = launch_demod_wave(a, nslots, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        CAPCHK(hipGetLastError());
        std::vector<int64_t> pos((size_t)nslots), len((size_t)nslots);
        std::vector<int32_t> nfr((size_t)nslots);
        CAPCHK(hipMemcpyAsync(pos.data(), w->d_posB, sizeof(int64_t) * nslots, hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(len.data(), w->d_consB, sizeof(int64_t) * nslots, hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(nfr.data(), w->d_nfB, sizeof(int32_t) * nslots, hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        for (int q = 0; q < nslots; q++) r.frames_demodulated += nfr[q] + (mode[q] == 1 ? F : 0);
        if (r.passes > 1) r.segments_rerun += nslots;
        // walk the chain by sample position: from the head through the replica of each next segment that starts where this one ended
        std::vector<int32_t> chain{0}, pq;
        for (int s2 = v + 1; s2 <= s_hi; s2++) {
            const int cur = chain.back();
            if (nfr[cur] != F) break;
            const int64_t end = pos[cur] + len[cur];
            int found = -1;
            for (int q = first_slot[s2]; q < first_slot[s2] + nslot_of[s2] && found < 0; q++) if (pos[q] == end) found = q;
            if (found < 0) break;
            pq.push_back(cur); pq.push_back(found);
            chain.push_back(found);
        }
        // ... and keep it as far as the states agree too
        int why = 0;
        if (!pq.empty()) {
            const int np = (int)pq.size() / 2;
            std::vector<int32_t> okv((size_t)np);
            CAP_H2D(w->d_ok, pq.data(), sizeof(int32_t) * pq.size(), st);
            hipLaunchKernelGGL(verify_kernel, dim3(np), dim3(kThreads), 0, st, state, snap, (const int64_t *)w->d_posB, (const int64_t *)w->d_consB,
                               (const int32_t *)w->d_ok, w->d_ok + pq.size(), Ndft, hist_elems, d.M,
                               d.est_band ? Ndft / 2 : 0, d.est_band ? Ndft / 2 + 16 * d.est_band : Ndft);
            CAPCHK(hipGetLastError());
            CAPCHK(hipMemcpyAsync(okv.data(), w->d_ok + pq.size(), sizeof(int32_t) * np, hipMemcpyDeviceToHost, st));
            CAPCHK(hipStreamSynchronize(st));
            for (int i = 0; i < np; i++) if (okv[i]) { why = okv[i]; chain.resize((size_t)i + 1); break; }
        }
        const int na = (int)chain.size(), nv = v + na;
        // what the slots measured: the verified chain's lengths are exact, the centre replicas' the best guess for the rest
        for (int s2 = v + 1; s2 <= s_hi; s2++) { const int qc = first_slot[s2] + nslot_of[s2] / 2; if (nfr[qc] == F) len_est[s2] = len[qc]; }
        for (int i = 0; i < na; i++) {
            len_est[v + i] = len[chain[i]];
            if (i) symbol_moves += std::abs(off_of[chain[i]] - off_of[chain[i - 1]]);
            from[v + i] = chain[i];
        }
        if (debug) {
            int omin = 0, omax = 0;
            for (int i = 0; i < na; i++) { omin = std::min(omin, off_of[chain[i]]); omax = std::max(omax, off_of[chain[i]]); }
            fprintf(stderr, "capture pass %d: head %d, %d slots over segments %d..%d (up to %d replicas), chain verified up to segment %d of %d through replicas %+d..%+d (verify code %d)%s\n",
                    r.passes, v, nslots, v, s_hi, nslot_of[s_hi], nv, S, omin, omax, why,
                    nv > s_hi || nfr[chain.back()] != F ? "" : why ? ": the replica at the chain's end is in another state" : ": no replica starts at the chain's end");
        }
        // rows of the newly verified segments to the caller's arrays; the chain's new end state
        CAP_H2D(w->d_from, from.data(), sizeof(int32_t) * S, st);
        hipLaunchKernelGGL(gather_kernel, dim3(na, 3), dim3(kThreads), 0, st, (const int32_t *)w->d_from, (const int32_t *)w->d_nfB, v, (int)F,
                           (int)fb, filt_floats, (const uint8_t *)w->r_bits, d_bits, (const float *)w->r_filt, d_rx_filt, (const float *)w->r_stats, stats);
        const int ql = chain.back();
        hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, KEEP, ql, Ndft, hist_elems);
        CAPCHK(hipGetLastError());
        chain_pos = pos[ql] + len[ql];
        if (nv == S || nfr[ql] < F) {                      // the last segment, or the one the samples (or the output rows) ran out in
            total_frames = (int64_t)(nv - 1) * F + nfr[ql];
            total_consumed = chain_pos;
            // a slot's row block ends after F frames: if the last segment filled it, samples and rows may be left
            more_after_last = nv == S && nfr[ql] == F && total_frames < max_frames;
            break;
        }
        v = nv;
    }
    const int final_slot = KEEP;
    if (more_after_last) {
        // the rest, sequentially, from the end state (a few frames: the capture was cut into S segments of F frames by its nominal length)
        skip_all();
        segB[0] = SegDesc{total_consumed, total_frames, (int32_t)std::min<int64_t>(max_frames - total_frames, 0x7fffffff), 0};
        CAP_H2D(w->d_segB, segB.data(), sizeof(SegDesc) * segB.size(), st);
        hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, 0, KEEP, Ndft, hist_elems);
        a.io = DemodIO{(const uint8_t *)d_in, 0, nsamp, d_bits, 0, d_rx_filt, 0, stats, 0, w->d_nfB, w->d_consB, F,
                       SoftOut{nullptr, 0, nullptr, 0, nullptr, 0}, w->d_segB};
        hipError_t e = launch_demod_wave(a, 1, st);
        if (e != hipSuccess) { h->last_hip = (int)e; return PIRIP_ERR_HIP; }
        hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, KEEP, 0, Ndft, hist_elems);
        int32_t nf = 0; int64_t c = 0;
        CAPCHK(hipMemcpyAsync(&nf, w->d_nfB, sizeof(nf), hipMemcpyDeviceToHost, st));
        CAPCHK(hipMemcpyAsync(&c, w->d_consB, sizeof(c), hipMemcpyDeviceToHost, st));
        CAPCHK(hipStreamSynchronize(st));
        total_frames += nf; total_consumed += c; r.frames_demodulated += nf;
    }
    // the capture's end state becomes the stream's (slot 0), with ppm recomputed in frame order
    hipLaunchKernelGGL(copy_state_kernel, dim3(1), dim3(kThreads), 0, st, state, 0, final_slot, Ndft, hist_elems);
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
