/* oracle/ldpc_independent.c -- TEST INFRASTRUCTURE ONLY (CPU checker; parity unpinned).
 *
 * A SECOND, independent CPU receiver for the FSK_LDPC row, written so that it shares NO arithmetic choice with
 * the product: oracle/ldpc_oracle.c mirrors the kernel (binary16 soft-bit exchange, 64-lane "wave order" frame sums, table
 * phi / ln I0) and can therefore only prove that the kernel computes what it was designed to compute. This file is what that
 * design is measured AGAINST (tools/ldpc_fer.py -> profiles/r04_ldpc_fer.txt, tests/test_ldpc.py):
 *   - soft bits stay float32, never rounded to binary16, never clamped;
 *   - every sum runs serially in index order (what codec2's scalar C does);
 *   - the decoder is the textbook flooding sum-product in double precision, phi(x) = -log(tanh(x/2)) evaluated with libm,
 *     no tables, at most max_iter iterations, stop when all checks are satisfied (/root/reference/README.md:211: "<= 15").
 * Two LLR mappings, selected at create:
 *   mode 1  RICIAN            the textbook non-coherent M-FSK log-likelihood ln I0(2 A r / sigma^2) with ln I0 evaluated exactly
 *                             (power series / asymptotic expansion in double), max-log bit metrics for 4-FSK
 *   mode 2  UPSTREAM-RECALLED codec2's fsk_rx_filt_to_llrs() as the author recalls it (mpdecode_core.c: FskDemod -> Somap ->
 *                             sign flip): symbol metric logbesseli0(2 * SNRest * |r| / v_est) with CML's piecewise-quadratic
 *                             logbesseli0, max_star0 = plain max, llr = -(num - den). NOT checkable here (codec2 is not under
 *                             /root/reference). The five logbesseli0 segments were checked numerically against ln I0
 *                             (tests/test_ldpc.py: within 0.0012 / 0.0018 / 0.011 / 0.031 of the true value on [0,1) / [1,2) /
 *                             [2,5) / [5,20), 0.061 on [20,60]), which a mis-remembered coefficient would not be.
 *                             codec2's decoder (phi0 look-up with its own break points) is NOT recalled well enough to restate and
 *                             is not imitated: mode 2 uses the same double-precision sum-product as mode 1.
 *   mode 3  mode 2's mapping, and the decoder's phi limited to the range codec2's phi0() covers, as recalled [UPSTREAM-RECALLED CML
 *                             MpDecode phi0: "if (x > 10) return 0; else if (x < 9.08e-5) return 10; ..."]: exact phi in between (the
 *                             staircase of the original is not recalled). Messages then saturate at 10 as the reference's do -- the
 *                             property tools/ldpc_precision.py shows to decide which marginal frames decode.
 * The unique-word search / sync state machine / CRC16 / record layout follow the recalled control flow of
 * freedv_rx_fsk_ldpc_data [UPSTREAM-RECALLED] and what the reference pins (tx/frame_repeater.c:55-62,71,80,88;
 * tx/rpitx_fsk.cpp:75-83,394-395), written here a second time from that description rather than shared with ldpc_oracle.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IND_UW 32
#define IND_SYNC 0x2
#define IND_BITS 0x4
#define IND_BIT_ERRORS 0x8

typedef struct {
    int n, k, m, nedges, max_iter, t1, t2, tbad, M, Nsym, Nbits, bpf, mode;
    uint8_t uw[IND_UW];
    int *chk_first, *chk_var;       /* check c owns edges chk_first[c] .. chk_first[c+1]-1, edge e touches variable chk_var[e] */
    float *win;                     /* sliding window: two frames of float32 soft bits, oldest first */
    int in_sync, uw_pos, misses, last_uw_err;
} LDPC_INDEP;

LDPC_INDEP *indep_ldpc_create(int n, int k, const int32_t *row_ptr, const int32_t *col_idx, const uint8_t *uw, int max_iter,
                              int uw_thresh1, int uw_thresh2, int bad_uw_thresh, int M, int Nsym, int mode)
{
    LDPC_INDEP *d = (LDPC_INDEP *)calloc(1, sizeof(*d));
    d->n = n; d->k = k; d->m = n - k; d->nedges = row_ptr[n - k]; d->max_iter = max_iter;
    d->t1 = uw_thresh1; d->t2 = uw_thresh2; d->tbad = bad_uw_thresh; d->M = M; d->Nsym = Nsym; d->mode = mode;
    d->Nbits = (M == 4) ? 2 * Nsym : Nsym;
    d->bpf = IND_UW + n;
    memcpy(d->uw, uw, IND_UW);
    d->chk_first = (int *)malloc(sizeof(int) * (size_t)(d->m + 1));
    d->chk_var = (int *)malloc(sizeof(int) * (size_t)d->nedges);
    for (int c = 0; c <= d->m; c++) d->chk_first[c] = row_ptr[c];
    for (int e = 0; e < d->nedges; e++) d->chk_var[e] = col_idx[e];
    d->win = (float *)calloc((size_t)2 * d->bpf, sizeof(float));
    return d;
}

void indep_ldpc_destroy(LDPC_INDEP *d)
{
    if (!d) return;
    free(d->chk_first); free(d->chk_var); free(d->win); free(d);
}

/* ln I0(x), x >= 0, to double accuracy: power series sum (x/2)^(2t) / (t!)^2 below 30, Hankel's asymptotic series above */
static double ln_bessel_i0(double x)
{
    if (x < 30.0) {
        const double q = 0.25 * x * x;
        double term = 1.0, sum = 1.0;
        for (int t = 1; t < 500; t++) {
            term *= q / ((double)t * (double)t);
            sum += term;
            if (term < 1e-18 * sum) break;
        }
        return log(sum);
    }
    /* I0(x) ~ e^x / sqrt(2 pi x) * (1 + 1/(8x) + 9/(128 x^2) + 225/(3072 x^3) + 11025/(98304 x^4)) */
    const double ix = 1.0 / x;
    const double s = 1.0 + ix * (0.125 + ix * (9.0 / 128.0 + ix * (225.0 / 3072.0 + ix * (11025.0 / 98304.0))));
    return x - 0.5 * log(2.0 * M_PI * x) + log(s);
}

/* [UPSTREAM-RECALLED] CML / codec2 mpdecode_core.c logbesseli0(): piecewise quadratic fit of ln I0 */
static float logbesseli0_recalled(float x)
{
    if (x < 1.0f) return 0.226f * x * x + 0.0125f * x - 0.0012f;
    if (x < 2.0f) return 0.1245f * x * x + 0.2177f * x - 0.108f;
    if (x < 5.0f) return 0.0288f * x * x + 0.6314f * x - 0.5645f;
    if (x < 20.0f) return 0.002f * x * x + 0.9048f * x - 1.2997f;
    return 0.9867f * x - 2.2053f;
}

/* for the test that pins the recalled coefficients to the function they approximate */
double indep_ln_i0(double x) { return ln_bessel_i0(x); }
float indep_logbesseli0_recalled(float x) { return logbesseli0_recalled(x); }

/* soft decisions of one demodulator call, fsk_demod_sd layout [m][sym] -> Nbits float32 LLRs, positive = bit 0 */
void indep_ldpc_llr(const LDPC_INDEP *d, const float *r, float *llr)
{
    const int M = d->M, ns = d->Nsym;
    /* the frame's signal and noise power the way fsk_demod_core accumulates them: serially, symbol by symbol, float32 */
    float sig = 0.0f, nse = 0.0f;
    for (int i = 0; i < ns; i++) {
        float total = 0.0f, best = 0.0f;
        for (int m = 0; m < M; m++) {
            const float p = r[m * ns + i] * r[m * ns + i];
            total += p;
            if (p > best) best = p;
        }
        sig += best;
        nse += (total - best) / (float)(M - 1);
    }
    sig = sig / (float)ns;
    nse = nse / (float)ns + 1e-12f;
    const float v2 = sig - nse;
    const float v_est = v2 > 0.0f ? sqrtf(v2) : 0.0f;
    const float snr_est = sig / nse;
    const int bps = (M == 4) ? 2 : 1;
    for (int i = 0; i < ns; i++) {
        double metric[4] = {0, 0, 0, 0};
        for (int m = 0; m < M; m++) {
            const float mag = fabsf(r[m * ns + i]);
            if (d->mode >= 2) {
                /* FskDemod(): y_envelope = sqrt(yr^2 / v_est^2); out = logbesseli0(2 * SNR * y_envelope) */
                const float env = v_est > 0.0f ? sqrtf((mag * mag) / (v_est * v_est)) : 0.0f;
                metric[m] = (double)logbesseli0_recalled((2.0f * snr_est) * env);
            } else {
                /* Rician envelope of the tone that carries the signal against Rayleigh envelopes of the others:
                 * log p(r | tone m sent) = ln I0(2 A r_m / sigma^2) + terms common to all m */
                metric[m] = ln_bessel_i0(2.0 * (double)v_est * (double)mag / (double)nse);
            }
        }
        /* Somap(): for every bit, best metric among the symbols with that bit = 1 (num) and = 0 (den), MSB first;
         * max_star0 is the plain max (max-log-MAP); fsk_rx_filt_to_llrs() returns -(num - den) */
        for (int b = 0; b < bps; b++) {
            const int mask = 1 << (bps - 1 - b);
            double num = -1e6, den = -1e6;
            for (int m = 0; m < M; m++) {
                if (m & mask) { if (metric[m] > num) num = metric[m]; }
                else          { if (metric[m] > den) den = metric[m]; }
            }
            llr[bps * i + b] = (float)(den - num);
        }
    }
}

static int g_phi_clip = 0;        /* set per decode call from the receiver's mode (3: the recalled phi0 range) */
static double phi_exact(double x)
{
    if (g_phi_clip) {
        if (x < 9.08e-5) return 10.0;
        if (x > 10.0) return 0.0;
    }
    /* -log(tanh(x/2)) = log1p(e^-x) - log1p(-e^-x), x > 0 */
    if (x < 1e-300) x = 1e-300;
    const double e = exp(-x);
    return log1p(e) - log1p(-e);
}

/* flooding sum-product in double; returns iterations used, *pcc = satisfied checks of the final hard decisions */
int indep_ldpc_decode(const LDPC_INDEP *d, const float *llr_in, uint8_t *hard, int *pcc)
{
    const int n = d->n, m = d->m, E = d->nedges;
    g_phi_clip = d->mode == 3;
    double *c2v = (double *)calloc((size_t)E, sizeof(double));      /* check -> variable messages */
    double *post = (double *)malloc(sizeof(double) * (size_t)n);    /* a-posteriori LLRs */
    double *mag = (double *)malloc(sizeof(double) * (size_t)E);
    for (int v = 0; v < n; v++) post[v] = (double)llr_in[v];
    int used = 0, ok = 0;
    for (int it = 1; it <= d->max_iter; it++) {
        /* check-node update from variable -> check messages post[v] - c2v[e] */
        for (int c = 0; c < m; c++) {
            const int a = d->chk_first[c], b = d->chk_first[c + 1];
            double total = 0.0;
            int parity = 0;
            for (int e = a; e < b; e++) {
                const double q = post[d->chk_var[e]] - c2v[e];
                mag[e] = phi_exact(fabs(q));
                total += mag[e];
                parity ^= (q < 0.0);
            }
            for (int e = a; e < b; e++) {
                const double q = post[d->chk_var[e]] - c2v[e];
                double out = phi_exact(total - mag[e]);
                if (out > 1e3) out = 1e3;
                mag[e] = (parity ^ (q < 0.0)) ? -out : out;
            }
        }
        /* variable-node update: a-posteriori = channel + all incoming */
        for (int v = 0; v < n; v++) post[v] = (double)llr_in[v];
        for (int e = 0; e < E; e++) { c2v[e] = mag[e]; post[d->chk_var[e]] += mag[e]; }
        for (int v = 0; v < n; v++) hard[v] = post[v] < 0.0;
        ok = 0;
        for (int c = 0; c < m; c++) {
            int x = 0;
            for (int e = d->chk_first[c]; e < d->chk_first[c + 1]; e++) x ^= hard[d->chk_var[e]];
            ok += (x == 0);
        }
        used = it;
        if (ok == m) break;
    }
    free(c2v); free(post); free(mag);
    *pcc = ok;
    return used;
}

/* CRC-16/CCITT-FALSE, bit by bit (poly 0x1021, init 0xFFFF, no reflection, no final xor) */
static unsigned crc16_bitwise(const uint8_t *p, int nbytes)
{
    unsigned crc = 0xFFFFu;
    for (int i = 0; i < nbytes; i++) {
        crc ^= (unsigned)p[i] << 8;
        for (int b = 0; b < 8; b++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) & 0xFFFFu : (crc << 1) & 0xFFFFu;
    }
    return crc;
}

static int uw_errors(const LDPC_INDEP *d, int pos)
{
    int e = 0;
    for (int u = 0; u < IND_UW; u++) e += (d->win[pos + u] < 0.0f) != (d->uw[u] != 0);
    return e;
}

/* one demodulator call: rx_filt (or NULL: a call without output) -> status byte; payload k/8 bytes; info[10] laid out like
 * oracle_ldpc_rx_call's so that the two receivers can be compared column by column */
int indep_ldpc_rx_call(LDPC_INDEP *d, const float *rx_filt, uint8_t *payload, int32_t *info)
{
    const int W = 2 * d->bpf, nb = d->Nbits;
    memmove(d->win, d->win + nb, sizeof(float) * (size_t)(W - nb));
    if (rx_filt) indep_ldpc_llr(d, rx_filt, d->win + W - nb);
    else memset(d->win + W - nb, 0, sizeof(float) * (size_t)nb);

    int sync = d->in_sync;
    if (!sync) {
        int best = 1 << 30, where = 0;
        for (int p = 0; p < d->bpf; p++) { const int e = uw_errors(d, p); if (e < best) { best = e; where = p; } }
        d->last_uw_err = best;
        if (best <= d->t1) { sync = 1; d->uw_pos = where; d->misses = 0; }
    } else {
        d->uw_pos -= nb;                                  /* the window slid by one call */
        if (d->uw_pos < 0) {                              /* the frame left the window: look for the next one a frame later */
            d->uw_pos += d->bpf;
            d->last_uw_err = uw_errors(d, d->uw_pos);
            if (d->last_uw_err > d->t2) { if (++d->misses >= d->tbad) sync = 0; }
            else d->misses = 0;
        }
    }
    int status = 0, iters = 0, pcc = 0, crc_ok = 0, where = -1, raw = 0;
    const int nbytes = d->k / 8;
    memset(payload, 0, (size_t)nbytes);
    if (sync) {
        status |= IND_SYNC;
        if (d->uw_pos >= 0 && d->uw_pos < nb) {            /* a whole frame has just become available */
            where = d->uw_pos;
            const float *cw = d->win + d->uw_pos + IND_UW;
            uint8_t *hard = (uint8_t *)malloc((size_t)d->n);
            iters = indep_ldpc_decode(d, cw, hard, &pcc);
            for (int v = 0; v < d->n; v++) raw += (cw[v] < 0.0f) != (hard[v] != 0);
            for (int v = 0; v < d->k; v++) payload[v >> 3] |= (uint8_t)(hard[v] << (7 - (v & 7)));
            crc_ok = crc16_bitwise(payload, nbytes - 2) == (((unsigned)payload[nbytes - 2] << 8) | payload[nbytes - 1]);
            if (crc_ok) status |= IND_BITS;
            if (pcc != d->m) status |= IND_BIT_ERRORS;
            free(hard);
        }
    }
    d->in_sync = sync;
    info[0] = sync; info[1] = d->uw_pos; info[2] = d->last_uw_err; info[3] = d->misses; info[4] = iters; info[5] = pcc;
    info[6] = where; info[7] = crc_ok; info[8] = raw; info[9] = 0;
    return status;
}

/* a whole stream of calls (rx_filt [ncalls][M*Nsym]) in one go, for the FER sweeps */
void indep_ldpc_rx_stream(LDPC_INDEP *d, const float *rx_filt, int ncalls, uint8_t *status, uint8_t *payload, int32_t *info)
{
    const int per = d->M * d->Nsym, nbytes = d->k / 8;
    for (int c = 0; c < ncalls; c++)
        status[c] = (uint8_t)indep_ldpc_rx_call(d, rx_filt + (size_t)c * per, payload + (size_t)c * nbytes, info + (size_t)c * 10);
}
