/* include/pirip_hip.h -- C-ABI of the MI355X-native FSK receive path (libpirip_hip.so).
 *
 * This is the drop-in boundary for pirip's IQ->bits hot path. pirip itself has no
 * plugin/operator API; the path sits behind (1) process boundaries -- the executables and
 * byte streams its scripts drive -- and (2) the libcodec2 / libcsdr C APIs that `rtl_fsk`
 * links (/root/reference/build_rtlsdr.sh:9). Both levels are served from this library:
 *
 *   section A  batch-of-streams device API (pirip_hip_*)  : what bench.py / a multi-channel
 *              receiver binds; device pointers in, device pointers out, explicit HIP stream.
 *   section B  csdr front end (pirip_hip_decim_*)          : convert_u8_f | fir_decimate_cc D
 *              | convert_f_s16 of /root/reference/README.md:109,162 as one device stage.
 *   section C  libcodec2-compatible single-stream shim      : fsk_create_hbr / fsk_nin /
 *              fsk_demod / fsk_demod_sd ... with codec2's own names and calling convention
 *              [UPSTREAM-RECALLED codec2 src/fsk.h], so `rtl_fsk` and `fsk_demod` link
 *              against libpirip_hip.so instead of libcodec2.so.
 *   section D  libcsdr-compatible entry points              : convert_u8_f, convert_f_s16,
 *              firdes_*, fir_decimate_cc
 *              [UPSTREAM-RECALLED csdr libcsdr.h].
 *
 * No torch types, no C++ types: plain pointers and sizes. Every function returns
 * PIRIP_OK (0) or a negative error; nothing here falls back to a CPU implementation --
 * without a usable HIP device every compute entry point returns PIRIP_ERR_NO_DEVICE.
 *
 * Reference call sites that pin the behaviour (arguments, byte formats):
 *   fsk_demod -d -p 24 2 240000 10000   /root/reference/test/loopback_rtl_sdr.sh:16
 *   fsk_demod --fsk_lower 500 --fsk_upper 25000 -d -p 24 ...  /root/reference/README.md:105
 *   csdr convert_u8_f | fir_decimate_cc 45 | convert_f_s16 | fsk_demod -c 2 40000 1000
 *                                        /root/reference/README.md:109
 *   rtl_fsk ... (in-process convert_u8_f + fsk_demod)  /root/reference/test/loopback_rtl_fsk.sh:10
 */
#ifndef PIRIP_HIP_H
#define PIRIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ----------------------------------------------------------------------------------- */
/* status codes                                                                         */
/* ----------------------------------------------------------------------------------- */
#define PIRIP_OK               0
#define PIRIP_ERR_BAD_ARG     (-1)   /* NULL pointer / size out of range                  */
#define PIRIP_ERR_BAD_CONFIG  (-2)   /* what codec2's fsk_create_core() would assert on   */
#define PIRIP_ERR_NO_DEVICE   (-3)   /* no HIP device / HIP runtime error at create       */
#define PIRIP_ERR_HIP         (-4)   /* HIP runtime error during a call                   */
#define PIRIP_ERR_NOMEM       (-5)
#define PIRIP_ERR_UNSUPPORTED (-6)

/* input sample formats (what the front end hands to the demodulator) */
#define PIRIP_IN_CU8_FSKDEMOD 0   /* fsk_demod -d : interleaved u8 IQ, (x-127)/128         */
#define PIRIP_IN_CU8_CSDR     1   /* csdr convert_u8_f / rtl_fsk : u8 IQ, x/127.5-1        */
#define PIRIP_IN_CS16         2   /* fsk_demod -c : interleaved s16 IQ, x/FDMDV_SCALE      */
#define PIRIP_IN_CF32         3   /* COMP {float real, imag}                               */

#define PIRIP_MODE_M_MAX 4
#define PIRIP_FSK_DEFAULT_P 8
#define PIRIP_FSK_DEFAULT_NSYM 50
#define PIRIP_FDMDV_SCALE 750
#define PIRIP_STATS_PER_FRAME 10  /* f_est[0..3], norm_rx_timing, SNRest, nin_next, ppm, rx_sig_pow, rx_nse_pow
                                     (the last two: what freedv_get_fsk_S_and_N() hands rtl_fsk -L, README.md:59) */

/* ----------------------------------------------------------------------------------- */
/* section A : batch-of-streams demodulator                                             */
/* ----------------------------------------------------------------------------------- */

/* Modem configuration: the arguments of codec2 fsk_create_hbr() + fsk_set_freq_est_limits()
 * + fsk_set_freq_est_alg() [UPSTREAM-RECALLED fsk.h], as driven by fsk_demod's argv
 * (/root/reference/README.md:105). */
typedef struct pirip_fsk_params {
    int Fs;             /* sample rate, Hz                                                */
    int Rs;             /* symbol rate; Fs % Rs == 0                                      */
    int M;              /* 2 or 4 tones                                                   */
    int P;              /* timing oversample (-p); (Fs/Rs) % P == 0, P >= 4               */
    int Nsym;           /* symbols per demod frame (default 50)                           */
    int est_min;        /* --fsk_lower, Hz                                                */
    int est_max;        /* --fsk_upper, Hz (est_min == est_max == 0: fsk_create defaults) */
    int freq_est_type;  /* 0 = peak picker, 1 = --mask comb                               */
    int tone_spacing;   /* --mask spacing, Hz (only read when freq_est_type == 1)         */
    int in_format;      /* PIRIP_IN_*                                                     */
} pirip_fsk_params;

/* Derived constants (what codec2 keeps in struct FSK). */
typedef struct pirip_fsk_info {
    int Ts, N, Nmem, Ndft, Nbits, nin_max, nstreams;
    int bytes_per_sample;   /* of the configured in_format (one complex sample)           */
} pirip_fsk_info;

typedef struct pirip_hip_demod pirip_hip_demod;   /* opaque: nstreams x struct FSK on device */

/* Create `nstreams` independent demodulators (one struct FSK each, all the same
 * configuration) resident on HIP device `device` (-1 = current device). Replaces
 * nstreams x fsk_create_hbr()+fsk_set_freq_est_limits()+fsk_set_freq_est_alg(). Every later call on
 * the handle runs on that device (the library selects it), whatever the caller's current device is. */
int pirip_hip_create(const pirip_fsk_params *params, int nstreams, int device, pirip_hip_demod **out);

/* Every constant of the demodulator that this repository holds FROM RECALL of codec2's fsk.c / fsk_demod.c rather than from a source it
 * could read (SURVEY.md 8c's verify-when-source-appears list; /root/reference/build_codec2.sh:3-5 clones an un-pinned HEAD, so any of
 * them may differ on the day the oracle is pinned). Each is one field here and in the CPU restatement (oracle/fsk_oracle.h:
 * fsk_oracle_recalled, same layout); pirip_hip_recalled_defaults() fills in today's values, which is what pirip_hip_create() runs.
 * The specialised kernels (PIRIP_KERNEL_WAVE / _BLOCK) are built around the defaults of the fields marked [k]: a handle created
 * with another value of one of those is served by the any-configuration kernel, which reads all of them from the plan. The others
 * are table / plan data for every kernel. oracle/pin_against_ref.py names, per failing case, the field whose other value repairs it. */
typedef struct pirip_fsk_recalled {
    int hann_denominator_ndft;  /* Hann window 0.5 - 0.5 cos(2 pi i / D): 0: D = Ndft - 1 (recalled), 1: D = Ndft                      */
    float tc;                   /* smoothing of Sf, 0.1                                                                                 */
    float est_space_rs;         /* blanking around a found peak, in symbol rates: 0.75                                                  */
    float nin_threshold;        /* [k] |norm_rx_timing| beyond which nin moves: 0.25                                                    */
    int nin_step_div;           /* [k] nin moves by Ts / nin_step_div samples: 4 (older fsk.c: 2)                                       */
    float s16_scale;            /* [k] FDMDV_SCALE, the divisor of `fsk_demod -c` int16 samples: 750 (codec2_fdmdv.h; 1000 elsewhere)   */
    float u8d_offset;           /* [k] fsk_demod -d: (x - u8d_offset) / u8d_scale: 127                                                  */
    float u8d_scale;            /* [k]                                              128                                                 */
    int ndft_rule;              /* [k] 0: bins of 0.1 Rs, next power of two (recalled); 1: largest power of two <= N (older fsk.c)      */
    int sf_power;               /* [k] what is smoothed into Sf: 0: |X| (recalled), 1: |X|^2                                            */
} pirip_fsk_recalled;
void pirip_hip_recalled_defaults(pirip_fsk_recalled *r);
/* pirip_hip_create with the recalled constants spelled out (recalled == NULL: the defaults). PIRIP_ERR_BAD_CONFIG for values no
 * demodulator can run (tc outside (0, 1], scales <= 0, nin_step_div < 2 or a step of 0 samples, a threshold outside (0, 0.5)). */
int pirip_hip_create_recalled(const pirip_fsk_params *params, const pirip_fsk_recalled *recalled, int nstreams, int device, pirip_hip_demod **out);
int pirip_hip_destroy(pirip_hip_demod *h);
int pirip_hip_get_info(const pirip_hip_demod *h, pirip_fsk_info *info);
/* Which device kernel serves this handle (chosen once at create): PIRIP_KERNEL_WAVE = a specialised wave-per-stream instance
 * (the reference's command-line shapes), PIRIP_KERNEL_GENERAL = the any-configuration kernel. Diagnostics: lets
 * a test or an operator confirm that a configuration is on the fast path. */
#define PIRIP_KERNEL_GENERAL 0
#define PIRIP_KERNEL_WAVE 2
#define PIRIP_KERNEL_EXACT 4    /* PIRIP_KERNEL=exact only: every frame in the CPU algorithm's own operation order (one thread walks the
                                 * oscillator and timing sums) -- bits, soft magnitudes, timing, nin, SNRest bit-equal to the CPU path at
                                 * any SNR; a cross-check at ~0.15 ms per frame and stream, never chosen by default */
#define PIRIP_KERNEL_BLOCK 3    /* workgroup-per-stream instance for long symbols (Ts = 240 / Ndft = 4096: rtl_fsk -r 1000 at 240 kS/s) */
int pirip_hip_get_kernel(const pirip_hip_demod *h);
/* The first frame of a stream after create / reset is demodulated, where the shape has P == Ts (`fsk_demod -p 24` at 24 samples per
 * symbol: a window of the integrator bank can then hold a single sample and the very first decision of a recording that starts one
 * sample before a symbol boundary is a float-rounding tie), by a prologue kernel that performs the CPU restatement's operations in its
 * order -- serial oscillator recursion, forward window sums, serial timing sum, glibc's atan2f -- so that frame is bit for bit the
 * oracle's (bits, soft magnitudes, timing). On by default; 0 switches it off (measurement A/B). Also PIRIP_EXACT0=0 at create. */
int pirip_hip_set_exact_first_frame(pirip_hip_demod *h, int enable);
/* The same as text: the instance (template arguments, streams per workgroup, waves per SIMD) or the general kernel with its
 * run-time shape -- what bench.py prints as config.kernel. */
int pirip_hip_get_kernel_name(const pirip_hip_demod *h, char *buf, size_t n);
/* Back to the state fsk_create_hbr() leaves (Sf = 0, oscillators at phase 0, nin = N). */
int pirip_hip_reset(pirip_hip_demod *h, void *hip_stream);
/* fsk_clear_estimators() for every stream [UPSTREAM-RECALLED codec2 fsk.c]: the smoothed spectrum Sf back to zero and nin back
 * to N -- oscillator phases, integrator memory, timing and ppm estimates stay, as upstream leaves them. */
int pirip_hip_clear_estimators(pirip_hip_demod *h, void *hip_stream);

/* Demodulate one batch. Stream s reads complex samples from
 *     (const char*)d_in + s*in_stride_bytes,  nsamp samples of the configured in_format,
 * and runs codec2's read loop on it: while nin samples remain (and frames < max_frames)
 * { fsk_demod(); advance by nin; nin = fsk_nin() }. State carries to the next call, so a
 * caller streams by re-presenting the unconsumed tail (see d_consumed) ahead of new data.
 * All d_* pointers are DEVICE pointers; outputs may be NULL when not wanted:
 *   d_bits    [s][frame][Nbits]  one bit per byte (fsk_demod's stdout format)
 *   d_rx_filt [s][frame][M*Nsym] soft magnitudes, fsk_demod_sd() layout [m][sym]
 *   d_stats   [s][frame][PIRIP_STATS_PER_FRAME]
 *   d_nframes [s] frames produced;  d_consumed [s] samples consumed (int64)
 * Strides are in elements of the respective array (bytes / floats / floats) per stream.
 * Work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream); the call does
 * not synchronise. */
int pirip_hip_demod_batch(pirip_hip_demod *h,
                          const void *d_in, size_t in_stride_bytes, int64_t nsamp,
                          uint8_t *d_bits, size_t bits_stride,
                          float *d_rx_filt, size_t filt_stride,
                          float *d_stats, size_t stats_stride,
                          int32_t *d_nframes, int64_t *d_consumed,
                          int64_t max_frames, void *hip_stream);

/* ONE long capture -- what `fsk_demod` gets when its input is a file ([UPSTREAM-RECALLED] fsk_demod.c usage: InputModemRawFile
 * OutputOneBitPerByteFile; the reference's own command lines give it pipes, /root/reference/README.md:105,109) -- demodulated on
 * many wavefronts with results identical to the read loop of pirip_hip_demod_batch on a one-stream handle: same frames, same bits / soft magnitudes / statistics rows, same d_consumed,
 * same state left behind (fsk_demod()'s frame-to-frame chain is cut into segments that are demodulated speculatively and kept only
 * where their start state proves, bit for bit, to be the state the segment before ended in; pirip_amd/csrc/capture.hip).
 * The handle's streams are the work slots: create it with nstreams = how many segments may run at once (>= 4; a few hundred to a
 * few thousand fill the GPU); stream slot 0 holds the capture's state between calls, so a capture too big for one call is
 * presented in pieces exactly like a stream (unconsumed tail ahead of the next piece). Handles served by the general kernel, short
 * inputs and nstreams < 4 take the sequential loop on slot 0 -- the results do not depend on the route.
 *   d_in       nsamp samples of the configured in_format (DEVICE pointer)
 *   d_bits     [frame][Nbits] (or packed, pirip_hip_set_bit_packing), d_rx_filt [frame][M*Nsym], d_stats [frame][PIRIP_STATS_PER_FRAME]:
 *              room for max_frames frames each; d_rx_filt and d_stats may be NULL
 *   nframes / consumed: frames produced and samples consumed (host)
 *   report     optional: how the call went
 * The call synchronises `hip_stream` (it decides on the host what to re-run). */
typedef struct pirip_capture_report {
    int32_t segments;            /* segments the capture was cut into (1: sequential route)                         */
    int32_t segment_frames;      /* frames per segment (0: sequential route)                                         */
    int32_t passes;              /* launches of the segment set: 1 = every speculative start verified at once        */
    int32_t segments_rerun;      /* segment runs in passes after the first                                           */
    int64_t frames_demodulated;  /* frames demodulated in all, warm-ups and re-runs included (>= nframes)            */
} pirip_capture_report;
int pirip_hip_demod_capture(pirip_hip_demod *h, const void *d_in, int64_t nsamp, uint8_t *d_bits, float *d_rx_filt, float *d_stats,
                            int64_t max_frames, int64_t *nframes, int64_t *consumed, pirip_capture_report *report, void *hip_stream);
/* The same with HOST buffers (upload, demodulate, download): what the fsk_demod tool calls when its input is a file. */
int pirip_hip_demod_capture_host(pirip_hip_demod *h, const void *in, int64_t nsamp, uint8_t *bits, float *rx_filt, float *stats,
                                 int64_t max_frames, int64_t *nframes, int64_t *consumed, pirip_capture_report *report);

/* Host-buffer convenience for one-stream callers (the CLI tools and section C): uploads
 * `nsamp` samples, runs stream 0, downloads. bits/rx_filt/stats sized for max_frames. */
int pirip_hip_demod_host(pirip_hip_demod *h, const void *in, int64_t nsamp,
                         uint8_t *bits, float *rx_filt, float *stats,
                         int64_t max_frames, int64_t *nframes, int64_t *consumed);
/* nin of stream 0 as of the last synchronised call (fsk_nin()). */
int pirip_hip_nin0(pirip_hip_demod *h);

/* Output format of d_bits / bits for subsequent calls: packed = 0 (default) one byte per bit, the
 * reference's stdout format; packed = 1: ceil(Nbits/8) bytes per frame, 8 bits per byte, MSB first --
 * codec2's freedv_pack order, the order `rpitx_fsk --packed` consumes (/root/reference/tx/rpitx_fsk.cpp:75-83,
 * script/frame_repeater:38). Strides and frame offsets are then in packed bytes. */
int pirip_hip_set_bit_packing(pirip_hip_demod *h, int packed);

/* Band-only frequency estimator (OPT-IN, default off). codec2's fsk_demod_freq_est smooths |X| of all Ndft FFT bins into Sf and
 * then searches the peaks in [est_min, est_max] only [UPSTREAM-RECALLED fsk.c]: bins outside that range cannot reach any output
 * of fsk_demod (bits, soft decisions, f_est, timing, SNR figures). With enable = 1 the demodulator computes and smooths only the
 * FFT bins the search can read -- today: Ndft = 256 handles on a wave instance built for it, peak estimator, 0 <= est_min: the 2-FSK
 * `fsk_demod -p 24` shape with est_max < 32 Fs/256 (bins 0 .. 31: up to 30 kHz at 240 kS/s) and the 4-FSK P = 8 shape with est_max <= 64 Fs/256
 * (bins 0 .. 63), both 8-bit input formats; PIRIP_ERR_UNSUPPORTED elsewhere. The surviving
 * bins go through the same butterflies on the same operands: Sf inside the band, f_est and every output stay bit-identical to the
 * full estimator's (tests/test_gpu_parity.py::test_band_only_estimator_*). What changes: Sf OUTSIDE the band is no longer updated
 * (pirip_hip_get_Sf returns stale values there -- leave it off where the whole spectrum is an output, as for rtl_fsk's dashboard),
 * pirip_hip_set_freq_est_limits() refuses a range that leaves the band, and switching it off again resets the streams. */
int pirip_hip_set_estimator_band_only(pirip_hip_demod *h, int enable);

/* fsk_enable_burst_mode() for every stream of the handle [UPSTREAM-RECALLED fsk.c]: nin is reset to N
 * and no longer follows the timing estimate. */
int pirip_hip_set_burst_mode(pirip_hip_demod *h, int enable);

/* Frequency-estimator spectrum of stream `s` (Ndft floats, DC at Ndft/2): the `SfdB`
 * source of rtl_fsk's dashboard JSON (/root/reference/script/dash.py:41). Synchronises. */
int pirip_hip_get_Sf(pirip_hip_demod *h, int s, float *Sf_host);
/* Eye diagram, the rx_eye / neyetr / neyesamp members of codec2's MODEM_STATS that fsk_demod_core fills for `fsk_demod -t`'s GUI
 * [UPSTREAM-RECALLED codec2 src/fsk.c, end of fsk_demod_core; src/modem_stats.h]: 8 / M traces per tone, each two symbols of
 * |f_int| (at most 160 points), trace i of tone m in row i*M + m. pirip_hip_enable_eye(h, 1) makes every later demodulator call
 * keep the traces of each stream's latest frame; it moves the handle to the any-configuration kernel (the only one that holds a
 * frame's integrator outputs) and RESETS the stream state, so call it right after pirip_hip_create. A diagnostic, as upstream's
 * is: not for throughput. pirip_hip_get_eye copies stream s's traces to rx_eye[8][160] (row stride 160), divided by their
 * largest value when `normalise` is set (upstream's default, fsk_stats_normalise_eye). Synchronises. */
int pirip_hip_enable_eye(pirip_hip_demod *h, int enable);
int pirip_hip_get_eye(pirip_hip_demod *h, int s, int normalise, float *rx_eye, int *neyetr, int *neyesamp);
/* Scalar state of stream `s` after the last call: f_est[0..3], norm_rx_timing, SNRest,
 * nin (as float), ppm -- the fields rtl_fsk reads out of struct FSK for its -v log line and
 * UDP JSON (/root/reference/script/dash.py:26-45). Synchronises. */
int pirip_hip_get_scalars(pirip_hip_demod *h, int s, float out8[8]);
/* All of it: what codec2 keeps in struct FSK between calls. snr_est / EbNodB / v_est are refreshed on OBSERVABLE frames --
 * frames whose per-frame stats are written, and the last frame of a call -- (the wave-per-stream kernel skips their
 * reductions elsewhere), so they equal codec2's values whenever d_stats is requested or calls carry one frame. */
typedef struct pirip_stream_state {
    int nin;
    float norm_rx_timing, ppm, snr_est, SNRest, EbNodB, v_est, f_est[4];
    float rx_sig_pow, rx_nse_pow;      /* mean power of the decided tone / of the other tones over the last observable frame */
} pirip_stream_state;
int pirip_hip_get_stream_state(pirip_hip_demod *h, int s, pirip_stream_state *out);
/* fsk_set_freq_est_limits() on a live handle: the search range changes, Sf / oscillators / timing state are kept. */
int pirip_hip_set_freq_est_limits(pirip_hip_demod *h, int est_min, int est_max);

/* ----------------------------------------------------------------------------------- */
/* section B : csdr front end  (convert_u8_f | fir_decimate_cc D [tbw] | convert_f_s16)  */
/* ----------------------------------------------------------------------------------- */
typedef struct pirip_hip_decim pirip_hip_decim;

/* decimation D, transition_bw (csdr default 0.05f -> int(4.0/0.05f) = 79 Hamming taps, padded to 80), cutoff 0.5/D.
 * out_s16 != 0 appends convert_f_s16 (interleaved s16 IQ out), else complex float out. */
int pirip_hip_decim_create(int decimation, float transition_bw, int out_s16, int device,
                           pirip_hip_decim **out);
int pirip_hip_decim_destroy(pirip_hip_decim *d);
int pirip_hip_decim_taps(const pirip_hip_decim *d, float *taps, int *ntaps);   /* host copy  */
/* Tap-loop arithmetic. 0 (default): one multiply and one add per tap and component in ascending tap order -- the scalar csdr loop's
 * float32 result bit for bit. OPT-IN measurement variants (also PIRIP_DECIM_FMA=1 / 2 at create): 1 = the same conversion, the
 * accumulation fused (acc = fma(y, h, acc)); 2 = the affine u8 map pulled out of the sum (acc = fma(byte, h, acc), y = acc/127.5 - sum h).
 * Upstream csdr is built -O3 -ffast-math [UPSTREAM-RECALLED]: which float32 result the shipped binary produces is a property of that
 * build's vectoriser, so neither variant is "wrong" a priori -- they are simply not what oracle/csdr_oracle.c states. */
#define PIRIP_DECIM_EXACT   0
#define PIRIP_DECIM_FMA     1
#define PIRIP_DECIM_FMA_RAW 2
int pirip_hip_decim_set_arith(pirip_hip_decim *d, int mode);
int pirip_hip_decim_get_arith(const pirip_hip_decim *d);
/* Number of outputs fir_decimate_cc yields for n_in inputs presented as ONE buffer:
 * floor((n_in - ntaps_padded)/D) + 1, or 0. */
int64_t pirip_hip_decim_nout(const pirip_hip_decim *d, int64_t n_in);
/* Stream s: u8 IQ at (const uint8_t*)d_in + s*in_stride_bytes (n_in complex samples) ->
 * out at (char*)d_out + s*out_stride_bytes, n_out = pirip_hip_decim_nout(n_in) samples. */
int pirip_hip_decim_batch(pirip_hip_decim *d, const uint8_t *d_in, size_t in_stride_bytes,
                          int64_t n_in, void *d_out, size_t out_stride_bytes, int nstreams,
                          void *hip_stream);

/* ----------------------------------------------------------------------------------- */
/* section B2 : synthetic Tx on the device                             */
/*   replaces the host pipeline fsk_get_test_bits | fsk_mod -c | (u8 quantiser, AWGN)   */
/*   used by /root/reference/README.md:101,142,218 to make test signals                 */
/* ----------------------------------------------------------------------------------- */
/* Stream s modulates nsym M-FSK symbols taken from d_bits + s*bits_stride (device, one bit per byte,
 * MSB first for 4FSK; bits_stride 0 = all streams send the same bits) on tones f1_hz[s] + m*tone_spacing_hz
 * (host array), drops skip_samples[s] leading samples (host array or NULL) and writes nsamp u8 IQ samples
 * (u8 = clamp(rintf(127 + amp*x)), x = 2*exp(j phase), i.e. fsk_mod -c output) to d_out + s*out_stride_bytes.
 * sigma > 0 adds sigma*N(0,1) per I and Q component before quantising (counter-based generator keyed by
 * seed, stream and sample). Phase is renormalised every PIRIP_FSK_DEFAULT_NSYM symbols like the fsk_mod tool.
 * Noise-free output is bit-identical to fsk_mod_c. Synchronous on hip_stream. */
int pirip_hip_synth_cu8(int Fs, int Rs, int M, int nstreams,
                        const int32_t *f1_hz, int tone_spacing_hz, const int32_t *skip_samples,
                        const uint8_t *d_bits, size_t bits_stride, int64_t nsym,
                        uint8_t *d_out, size_t out_stride_bytes, int64_t nsamp,
                        float amp, float sigma, uint64_t seed, void *hip_stream);

/* ----------------------------------------------------------------------------------- */
/* section E : FSK_LDPC receive (BASELINE config 4's second half)       */
/*   what `rtl_fsk --code NAME [-b]` does after fsk_demod_sd(): soft decisions -> LLRs ->  */
/*   32-bit unique-word sync -> LDPC sum-product decode (<= 15 iterations,                 */
/*   /root/reference/README.md:200-212) -> CRC16 -> payload bytes + rx_status, one record  */
/*   per demodulator call (/root/reference/tx/frame_repeater.c:55-62,71,80,88).            */
/*   [UPSTREAM-RECALLED codec2 freedv_fsk.c: freedv_rx_fsk_ldpc_data, mpdecode_core.c]     */
/*   The parity-check matrix, unique word and sync thresholds come from a code FILE        */
/*   (format: pirip_amd/csrc/fsk_ldpc.hpp); codec2's H_256_512_4 is not in /root/reference, */
/*   pirip_amd/data/standin_256_512_4.code is a labelled stand-in of the same shape.        */
/* ----------------------------------------------------------------------------------- */
#define PIRIP_RX_TRIAL_SYNC 0x1   /* rx_status bits [UPSTREAM-RECALLED codec2 freedv_api.h FREEDV_RX_*] */
#define PIRIP_RX_SYNC       0x2
#define PIRIP_RX_BITS       0x4   /* a frame was decoded and its CRC16 matches                          */
#define PIRIP_RX_BIT_ERRORS 0x8   /* not every parity check was satisfied                               */
#define PIRIP_LDPC_INFO_PER_CALL 10 /* state, uw_loc, uw_err, bad_uw, iter, pcc, frame bit position (-1 none), crc_ok,
                                       eraw (channel hard decisions the decoder changed), reserved */

typedef struct pirip_hip_ldpc pirip_hip_ldpc;
typedef struct pirip_ldpc_info {
    int n, k, bits_per_frame, data_bytes, nbits_per_call, max_iter, nstreams;
    char name[64];
} pirip_ldpc_info;

/* nstreams independent receivers for M-FSK with Nsym symbols per demodulator call (Nsym*log2(M) soft bits per call). */
int pirip_hip_ldpc_create(const char *code_path, int M, int Nsym, int nstreams, int device, pirip_hip_ldpc **out);
int pirip_hip_ldpc_destroy(pirip_hip_ldpc *h);
int pirip_hip_ldpc_get_info(const pirip_hip_ldpc *h, pirip_ldpc_info *info);
int pirip_hip_ldpc_reset(pirip_hip_ldpc *h, void *hip_stream);
/* Stream s consumes `ncalls` demodulator frames of soft decisions d_rx_filt + s*filt_stride (floats; each frame is
 * M*Nsym magnitudes in fsk_demod_sd() layout [m][sym] = pirip_hip_demod_batch's d_rx_filt); d_ncalls[s] (or NULL = all)
 * says how many of them are valid (d_nframes of the demodulator): the receiver advances by exactly that many calls -- the
 * rest of the batch is not demodulator output, gets status 0 / zero payload / info -1 and leaves no trace in the state that
 * carries to the next batch. Per (stream, call) it writes
 *   d_status  [s][ncalls]                 rx_status byte (PIRIP_RX_*)
 *   d_payload [s][ncalls][k/8]            packed payload bytes, zeros when the call produced no frame
 *   d_info    [s][ncalls][PIRIP_LDPC_INFO_PER_CALL]
 * i.e. the `rtl_fsk -b` record stream. Sync state and the last two frames of soft bits carry to the next call. */
int pirip_hip_ldpc_rx_batch(pirip_hip_ldpc *h, const float *d_rx_filt, size_t filt_stride, const int32_t *d_ncalls, int ncalls,
                            uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, void *hip_stream);
/* The whole receive chain of one batch in one call -- what `rtl_fsk --code NAME` does per block of IQ: stream s of `dem` (created
 * for the same nstreams / M / Nsym / device) demodulates its samples as pirip_hip_demod_batch would and its frames go straight
 * into receiver s of `h`: d_status / d_payload / d_info hold one record per demodulator call, [s][max_frames][...], calls
 * beyond d_nframes[s] as described above; d_stats (optional) the demodulator's per-frame statistics. Where the demodulator
 * kernel of the shape can, it writes the bit LLRs and their hard decisions itself (no soft magnitudes and no
 * LLR pass through HBM); otherwise the call is pirip_hip_demod_batch + pirip_hip_ldpc_rx_batch over an internal buffer. The
 * records are the same either way. pirip_hip_fsk_ldpc_last_path: 1 if the last such call took the fused hand-over, else 0.
 * From 4096 streams on (PIRIP_CHAIN_SPLIT_MIN=<n> moves the threshold, 0 = never) the fused form runs the batch as two ranges of
 * streams (5/8 and 3/8) on two internal HIP streams, forked from and joined back into `hip_stream`: the first range's LDS-bound
 * decode then runs beside the second range's VALU-bound demodulator. Same kernels on the same per-stream data: same records. */
int pirip_hip_fsk_ldpc_rx_batch(pirip_hip_demod *dem, pirip_hip_ldpc *h, const void *d_in, size_t in_stride_bytes, int64_t nsamp,
                                uint8_t *d_status, uint8_t *d_payload, int32_t *d_info, float *d_stats, size_t stats_stride,
                                int32_t *d_nframes, int64_t *d_consumed, int64_t max_frames, void *hip_stream);
int pirip_hip_fsk_ldpc_last_path(const pirip_hip_ldpc *h);
/* The same for several GROUPS of streams at once, each group with its own pair of handles and its own buffers: group g's call runs
 * on an internal HIP stream of its own (all but the last group at high priority), started behind everything queued on `hip_stream`
 * and joined back into it. The FSK_LDPC stages are bound by the LDS pipe and the demodulator by VALU issue, so group g's decode runs
 * beside group g+1's demodulator instead of after it (two groups, the first the bigger: 26.9 -> 25.4 ms per 8192 x 600 k samples at
 * 3.5 dB, tools/chain_overlap.py); the records of every group are what its own pirip_hip_fsk_ldpc_rx_batch call writes. All groups
 * share in_stride_bytes / nsamp / stats_stride / max_frames and must live on one device. */
typedef struct pirip_chain_group {
    pirip_hip_demod *dem; pirip_hip_ldpc *ldpc;
    const void *d_in; uint8_t *d_status; uint8_t *d_payload; int32_t *d_info; float *d_stats; int32_t *d_nframes; int64_t *d_consumed;
} pirip_chain_group;
int pirip_hip_fsk_ldpc_rx_batch_groups(const pirip_chain_group *groups, int ngroups, size_t in_stride_bytes, int64_t nsamp, size_t stats_stride,
                                       int64_t max_frames, void *hip_stream);
/* host-buffer convenience for a one-stream handle (rtl_fsk) */
int pirip_hip_ldpc_rx_host(pirip_hip_ldpc *h, const float *rx_filt, int ncalls, uint8_t *status, uint8_t *payload, int32_t *info);
/* the two numerical stages on their own (device pointers): bit LLRs of ncalls demodulator frames ([ncalls][Nbits]), and
 * the decoder on ncw codewords of LLRs ([ncw][n] -> hard codeword bits [ncw][n], {iterations, parity checks ok} [ncw][2]) */
int pirip_hip_ldpc_llr(pirip_hip_ldpc *h, const float *d_rx_filt, int ncalls, float *d_llr, void *hip_stream);
int pirip_hip_ldpc_decode_llr(pirip_hip_ldpc *h, const float *d_llr, int ncw, uint8_t *d_bits, int32_t *d_iter_pcc, void *hip_stream);

/* ----------------------------------------------------------------------------------- */
/* section C : libcodec2-compatible single-stream API (host buffers)                    */
/*             names and signatures as codec2 src/fsk.h [UPSTREAM-RECALLED]              */
/* ----------------------------------------------------------------------------------- */
#ifndef PIRIP_NO_CODEC2_SHIM
typedef struct { float real; float imag; } COMP;
#define MODE_M_MAX 4

/* struct FSK: the fields codec2's own programs read directly (fsk_demod.c, rtl_fsk.c: fsk->Nbits, fsk->Ndft, fsk->f_est[],
 * fsk->Sf[], fsk->norm_rx_timing, fsk->SNRest, fsk->ppm ...) are PUBLIC here, in codec2's order and with codec2's names
 * [UPSTREAM-RECALLED codec2 src/fsk.h], and are refreshed after every fsk_demod()/fsk_demod_sd() (Sf is downloaded on
 * demand through fsk->Sf: a host copy the library owns). Fields that have no host-side meaning in this build are kept for
 * source compatibility and left NULL/zero (hann_table, f_dc, fft_cfg, phi_c). Code must be recompiled against this header:
 * the layout follows the recalled upstream order but codec2 is not in /root/reference to check it against. The demodulator
 * state itself lives on the GPU behind `pirip_priv`. */
struct MODEM_STATS;
struct FSK {
    /* static parameters set up by fsk_create_hbr */
    int Ndft, Fs, N, Rs, Ts, Nmem, P, Nsym, Nbits, f1_tx, tone_spacing, mode;
    float tc;
    int est_min, est_max, est_space;
    float *hann_table;                 /* NULL: the window lives on the device */
    /* parameters used by the demodulator */
    float *Sf;                         /* [Ndft] smoothed magnitude spectrum, host copy refreshed by fsk_demod*() */
    COMP phi_c[MODE_M_MAX];            /* not mirrored (oscillators are device-side phase accumulators) */
    COMP *f_dc;                        /* NULL */
    void *fft_cfg;                     /* NULL */
    float norm_rx_timing;
    COMP tx_phase_c;                   /* modulator phase (fsk_mod / fsk_mod_c run on the CPU) */
    /* statistics generated by the demodulator */
    float EbNodB;
    float f_est[MODE_M_MAX];           /* peak-method tone estimates, Hz */
    float f2_est[MODE_M_MAX];          /* mask-method tone estimates, Hz (equal to f_est when the peak method drives the demod) */
    int freq_est_type;
    float ppm, SNRest, v_est, rx_sig_pow, rx_nse_pow;
    /* parameters used by mod/demod and the driving code */
    int nin, burst_mode, lock_nin;
    struct MODEM_STATS *stats;
    int normalise_eye;
    void *pirip_priv;                  /* library-private: device handle, staging buffers */
};

struct FSK *fsk_create(int Fs, int Rs, int M, int tx_f1, int tx_fs);
struct FSK *fsk_create_hbr(int Fs, int Rs, int M, int P, int Nsym, int f1_tx, int tone_spacing);
void fsk_destroy(struct FSK *fsk);
void fsk_set_freq_est_limits(struct FSK *fsk, int est_min, int est_max);
void fsk_set_freq_est_alg(struct FSK *fsk, int est_type);
uint32_t fsk_nin(struct FSK *fsk);
void fsk_demod(struct FSK *fsk, uint8_t rx_bits[], COMP fsk_in[]);
void fsk_demod_sd(struct FSK *fsk, float rx_filt[], COMP fsk_in[]);
void fsk_clear_estimators(struct FSK *fsk);
void fsk_enable_burst_mode(struct FSK *fsk);
/* Demod statistics, codec2's layout [UPSTREAM-RECALLED codec2 src/modem_stats.h]. The FSK demodulator fills Nc, snr_est
 * (the smoothed EbNodB, as upstream), foff, rx_timing, clock_offset, f_est and the eye diagram of the latest frame (rx_eye,
 * neyetr, neyesamp). The eye is OPT-IN here: a section C handle runs on its specialised kernel and leaves neyetr = 0 until
 * the program calls fsk_stats_normalise_eye() (either value) or the environment holds PIRIP_SHIM_EYE=1 -- from then on it runs
 * with pirip_hip_enable_eye (any-configuration kernel; asked for after the first fsk_demod() the demodulator state restarts,
 * with a note on stderr). The scatter (rx_symbols) / FFT members exist so that
 * code written against codec2 compiles and indexes them, and stay zero -- the FSK demodulator does not fill them upstream either. */
#define MODEM_STATS_NC_MAX      50
#define MODEM_STATS_NR_MAX      160
#define MODEM_STATS_ET_MAX      8
#define MODEM_STATS_EYE_IND_MAX 160
#define MODEM_STATS_NSPEC       512
#define MODEM_STATS_MAX_F_HZ    4000
#define MODEM_STATS_MAX_F_EST   4
struct MODEM_STATS {
    int Nc;
    float snr_est;
    COMP rx_symbols[MODEM_STATS_NR_MAX][MODEM_STATS_NC_MAX + 1];
    int nr, sync;
    float foff, rx_timing, clock_offset, sync_metric;
    int pre, post, uw_fails;
    float rx_eye[MODEM_STATS_ET_MAX][MODEM_STATS_EYE_IND_MAX];
    int neyetr, neyesamp;
    float f_est[MODEM_STATS_MAX_F_EST];
    float fft_buf[2 * MODEM_STATS_NSPEC];
    void *fft_cfg;
};
void fsk_get_demod_stats(struct FSK *fsk, struct MODEM_STATS *stats);
void fsk_stats_normalise_eye(struct FSK *fsk, int normalise_enable);   /* default on, as upstream */
void fsk_mod(struct FSK *fsk, float fsk_out[], uint8_t tx_bits[], int nbits);      /* CPU: Tx side */
void fsk_mod_c(struct FSK *fsk, COMP fsk_out[], uint8_t tx_bits[], int nbits);     /* CPU: Tx side */
/* accessors (kept from round 1; the public fields above carry the same values) */
int fsk_get_Nbits(struct FSK *fsk);
int fsk_get_Nsym(struct FSK *fsk);
int fsk_get_N(struct FSK *fsk);
int fsk_get_Ts(struct FSK *fsk);
int fsk_get_Ndft(struct FSK *fsk);
float fsk_get_norm_rx_timing(struct FSK *fsk);
float fsk_get_SNRest(struct FSK *fsk);
void fsk_get_f_est(struct FSK *fsk, float f_est[/*M*/]);
void fsk_get_Sf(struct FSK *fsk, float Sf[/*Ndft*/]);
#endif


/* ----------------------------------------------------------------------------------- */
/* section F : the FreeDV API calls `rtl_fsk --code` and `rpitx_fsk --code` bind          */
/*             [UPSTREAM-RECALLED codec2 src/freedv_api.h, freedv_fsk.c]                  */
/*   What upstream's rtl_fsk.c does in coded mode is libcodec2's FreeDV API in              */
/*   FREEDV_MODE_FSK_LDPC: freedv_open_advanced / freedv_nin / freedv_rawdatacomprx /       */
/*   freedv_get_rx_status / freedv_get_bits_per_modem_frame / freedv_close. The shape of    */
/*   that API is in the reference itself: struct freedv_advanced members Rs, Fs, M,         */
/*   codename (/root/reference/tx/rpitx_fsk.cpp:165,222,319-322), the Tx-side helpers it    */
/*   declares by hand (:33-40) and uses (:75-83,324-325,395,443,474), the rx_status bits    */
/*   (/root/reference/tx/frame_repeater.c:71,80,88). Each entry is a thin shim over         */
/*   sections C and E: one handle = one stream, caller owns every buffer, no error codes    */
/*   (NULL from open, as upstream).                                                          */
/*   codename -> code file: a path, else $PIRIP_CODE_DIR/<codename>.code, else               */
/*   <dir of libpirip_hip.so>/../data/<codename>.code (rtl_fsk --code's rule). codec2's      */
/*   H_256_512_4 is not part of this build: open returns NULL with a note unless a file of   */
/*   that name has been dropped in.                                                           */
/* ----------------------------------------------------------------------------------- */
#ifndef PIRIP_NO_CODEC2_SHIM
#define FREEDV_MODE_FSK_LDPC 9
#define FREEDV_RX_TRIAL_SYNC 0x1
#define FREEDV_RX_SYNC       0x2
#define FREEDV_RX_BITS       0x4
#define FREEDV_RX_BIT_ERRORS 0x8
struct freedv;
struct freedv_advanced {
    int interleave_frames;     /* unused, kept for layout */
    int M;                     /* 2 or 4 */
    int Rs;                    /* symbol rate, Hz */
    int Fs;                    /* sample rate, Hz */
    int first_tone;            /* Tx only; rpitx_fsk leaves it unset: any value is accepted */
    int tone_spacing;          /* Tx, and the comb of the mask estimator */
    char *codename;            /* LDPC code name, see above */
};
struct freedv *freedv_open_advanced(int mode, struct freedv_advanced *adv);   /* mode must be FREEDV_MODE_FSK_LDPC */
void freedv_close(struct freedv *f);
int freedv_nin(struct freedv *f);                          /* samples the next freedv_rawdatacomprx() reads */
int freedv_get_n_max_modem_samples(struct freedv *f);     /* upper bound of freedv_nin() */
/* one demodulator call: nin complex float samples in; returns the number of payload bytes written to packed_payload_bits --
 * freedv_get_bits_per_modem_frame()/8 when a frame with a good CRC16 came out of this call (FREEDV_RX_BITS), else 0 */
int freedv_rawdatacomprx(struct freedv *f, unsigned char *packed_payload_bits, COMP demod_in[]);
int freedv_get_rx_status(struct freedv *f);                /* FREEDV_RX_* of the last call */
int freedv_get_bits_per_modem_frame(struct freedv *f);    /* data bits per frame, CRC16 included (256 for a (512,256) code) */
void freedv_set_frames_per_burst(struct freedv *f, int framesperburst);   /* Tx-side burst length; stored */
void freedv_set_verbose(struct freedv *f, int verbosity);  /* >= 2: one line per decoded frame on stderr, README.md:200-208's columns */
void freedv_set_test_frames(struct freedv *f, int test_frames);            /* the ecdd column of that line counts payload bit errors */
struct FSK *freedv_get_fsk(struct freedv *f);              /* the demodulator: fsk_set_freq_est_limits / _alg, f_est[], Sf[] ... */
int freedv_get_sync(struct freedv *f);                     /* 1 while the receiver holds frame sync (FREEDV_RX_SYNC of the last call) */
void freedv_get_modem_stats(struct freedv *f, int *sync, float *snr_est);
void freedv_get_modem_extended_stats(struct freedv *f, struct MODEM_STATS *stats);   /* fsk_get_demod_stats + sync */
/* Tx-side helpers "not normally exposed by the FreeDV API" that rpitx_fsk declares itself (tx/rpitx_fsk.cpp:33-40); CPU */
int freedv_tx_fsk_ldpc_bits_per_frame(struct freedv *f);   /* 32 + n */
void freedv_tx_fsk_ldpc_framer(struct freedv *f, uint8_t frame[], uint8_t payload_data[]);   /* UW + data + parity, one bit per byte */
unsigned short freedv_gen_crc16(unsigned char *data_p, int length);
void freedv_pack(unsigned char *bytes, unsigned char *bits, int nbits);
void freedv_unpack(unsigned char *bits, unsigned char *bytes, int nbits);
/* --testframes payload: upstream's generator constants are not in the reference; this is the repo's own sequence, usable Tx + Rx together */
void ofdm_generate_payload_data_bits(uint8_t payload_data_bits[], int n);
#endif
/* codename -> code file by the rule above; 1 and the path in buf when found */
int pirip_hip_find_code(const char *codename, char *buf, size_t n);

/* ----------------------------------------------------------------------------------- */
/* section D : libcsdr-compatible entry points (host buffers) [UPSTREAM-RECALLED libcsdr.h] */
/* ----------------------------------------------------------------------------------- */
#ifndef PIRIP_NO_CSDR_SHIM
typedef struct { float i; float q; } complexf;
/* Element-wise format hops of the three-process pipe (/root/reference/README.md:109), run as
 * device kernels over host buffers (upload, convert, download) -- there is no host loop. In a
 * fused receiver use section B instead, which folds both into the decimator. */
void convert_u8_f(unsigned char *input, float *output, int length);
void convert_f_s16(float *input, short *output, int length);
int  firdes_filter_len(float transition_bw);
typedef enum window_s { WINDOW_BOXCAR, WINDOW_BLACKMAN, WINDOW_HAMMING } window_t;   /* [UPSTREAM-RECALLED libcsdr.h] */
#define WINDOW_DEFAULT WINDOW_HAMMING
void firdes_lowpass_f(float *output, int length, float cutoff_rate, window_t window);
void firdes_lowpass_f_hamming(float *output, int length, float cutoff_rate); /* = firdes_lowpass_f(..., WINDOW_HAMMING) */
/* Direct-form decimating FIR on the GPU; same contract as csdr: returns outputs written,
 * consumed input = returned * decimation. Input/output are host buffers of complex float. */
int  fir_decimate_cc(complexf *input, complexf *output, int input_size, int decimation,
                     float *taps, int taps_length);
#endif

/* ----------------------------------------------------------------------------------- */
/* misc                                                                                 */
/* ----------------------------------------------------------------------------------- */
const char *pirip_hip_version(void);        /* "pirip_hip 0.2 (gfx950)": 0.2 = PIRIP_STATS_PER_FRAME 10, pirip_stream_state with rx_sig_pow / rx_nse_pow */
/* ABI check for callers that were compiled earlier than the library they are linked against (the CMake relink of INTEGRATION.md):
 * compare with the PIRIP_HIP_ABI_VERSION / PIRIP_STATS_PER_FRAME / sizeof(pirip_stream_state) of the header the caller was built with
 * and refuse to run on a mismatch -- the library writes stats_per_frame floats per frame and a stream state of that many bytes. */
#define PIRIP_HIP_ABI_VERSION 2
int pirip_hip_abi(int *abi_version, int *stats_per_frame, size_t *stream_state_bytes);
/* 1 when the arguments -- the caller's compile-time PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME and sizeof(pirip_stream_state) -- are
 * this library's; call once at start-up: pirip_hip_abi_check(PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME, sizeof(pirip_stream_state)) */
int pirip_hip_abi_check(int abi_version, int stats_per_frame, size_t stream_state_bytes);
/* 16 hex digits of the sha256 of the wave demodulator kernels' gfx950 code object: identifies the kernel build a measurement file
 * (profiles/hbm_traffic.json) was taken on */
const char *pirip_hip_kernel_source_hash(void);
const char *pirip_hip_strerror(int status);
int pirip_hip_device_count(void);          /* 0 when no usable HIP device                  */
/* Device self-test of the estimator's correctly rounded square roots (the |X| in Sf = Sf (1-tc) + |X| tc,
 * [UPSTREAM-RECALLED codec2 fsk.c: fsk_demod_freq_est] uses sqrtf): both device variants against (float)sqrt((double)x) for
 * x = 0 and every float in [2^-96, FLT_MAX]; *mismatches = (v_sqrt variant's count << 32) | rsq variant's count, 0 on a
 * device where the kernels' fast path is valid. ~1 s. */
int pirip_hip_selftest_sqrt(uint64_t *mismatches);
/* Device self-test of the fused FSK_LDPC hand-over's divisions by constants (the frame's sums / Nsym = 50, the other tones' power / 3:
 * x * RN(1/c) corrected by one residual step instead of the 11-instruction IEEE quotient) against the device's own x / c for x = 0 and
 * every float in [2^-125, FLT_MAX]; *mismatches = (count for / 50 << 32) | count for / 3, 0 where the kernels' quick path is valid. ~1 s. */
int pirip_hip_selftest_div(uint64_t *mismatches);
/* The exact first frame's fine-timing angle is libm's atan2f restated in device code (fdlibm's float algorithm, which glibc ships):
 * this evaluates that restatement on device arrays so that a test can compare it with the host's atan2f bit for bit. */
int pirip_hip_selftest_atan2(const float *d_y, const float *d_x, float *d_out, int n);

#ifdef __cplusplus
}
#endif
#endif /* PIRIP_HIP_H */
