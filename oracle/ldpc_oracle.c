/* oracle/ldpc_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned).
 *
 * CPU restatement of the FSK_LDPC receive chain the HIP path implements (pirip_amd/csrc/ldpc_kernels.hip):
 * soft decisions -> bit LLRs -> unique-word sync state machine -> sum-product LDPC decode -> CRC16 -> payload + rx_status.
 * What the reference pins is the framing and the record/flag protocol (/root/reference/tx/rpitx_fsk.cpp:75-83,313-336,
 * 382-395,427-509; tx/frame_repeater.c:55-62,71,80,88; README.md:176-212); the decoder arithmetic, the LLR mapping, the
 * unique word and H_256_512_4 live in codec2 (freedv_fsk.c, mpdecode_core.c, H_256_512_4.c), which is NOT under
 * /root/reference and cannot be built here -> "parity unpinned": this file defines the arithmetic the GPU is tested
 * against, written from the published algorithms (Gallager's sum-product in the phi domain, Rician log-likelihoods of
 * non-coherent M-FSK) and the recalled control flow of freedv_rx_fsk_ldpc_data [UPSTREAM-RECALLED]. The parity-check
 * matrix and unique word arrive as arrays (the tests parse the same code file the product loads).
 *
 * Soft bits are EXCHANGED as IEEE binary16 (round to nearest even): the LLR stage rounds what it hands over, the decoder rounds
 * what it is given (idempotent on the receiver's own LLRs). |LLR| <= 24 and the decoder's phi table resolves 1/32 of an
 * octave, so half precision (2^-11 relative) is far below the decoder's own quantisation; it halves the only intermediate
 * the demodulator and the decoder pass through memory. This repo's definition, like the rest of this file.
 *
 * Plain scalar C, built -ffp-contract=off; sums run in index order except the two per-frame sums of the LLR stage, which run in the
 * receiver's defined wave order (wave_order_sum below: what a 64-lane reduction tree computes, restated serially).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LNI0_N 256
#define PHI_LO_EXP (-14)
#define PHI_HI_EXP 4
#define PHI_X_LO 9.08e-5f      /* [UPSTREAM-RECALLED codec2 phi0(): x < 9.08e-5 -> 10, x > 10 -> 0]: the product's range since round 5 */
#define PHI_X_HI 10.0f
#define PHI_STEPS 32
#define PHI_N ((PHI_HI_EXP - PHI_LO_EXP) * PHI_STEPS)
#define LLR_MAX 24.0f
#define LLR_MAX_UPSTREAM 1000.0f
#define LLR_UPSTREAM 0
#define LLR_RICIAN 1
#define RX_SYNC 0x2
#define RX_BITS 0x4
#define RX_BIT_ERRORS 0x8
#define UW_BITS 32

typedef struct {
    int n, k, m, E, max_iter, uw_thresh1, uw_thresh2, bad_uw_thresh, M, Nsym, Nbits, bpf;
    int llr_map;           /* LLR_UPSTREAM: codec2's fsk_rx_filt_to_llrs as recalled (the product's default); LLR_RICIAN: ln I0(2 A r / sigma^2) */
    uint8_t uw[UW_BITS];
    int32_t *row_ptr, *col_idx, *col_ptr, *col_edge;
    float lnI0[LNI0_N + 2], phi[PHI_N];
    /* receiver state */
    float *llr2;           /* two frames of soft bits, newest at the end */
    int state, loc, bad_uw, uw_err;
} LDPC_ORACLE;

LDPC_ORACLE *oracle_ldpc_create(int n, int k, const int32_t *row_ptr, const int32_t *col_idx, const uint8_t *uw, int max_iter,
                                int uw_thresh1, int uw_thresh2, int bad_uw_thresh, int M, int Nsym, int llr_map)
{
    LDPC_ORACLE *o = (LDPC_ORACLE *)calloc(1, sizeof(*o));
    o->llr_map = llr_map;
    o->n = n; o->k = k; o->m = n - k; o->E = row_ptr[n - k]; o->max_iter = max_iter;
    o->uw_thresh1 = uw_thresh1; o->uw_thresh2 = uw_thresh2; o->bad_uw_thresh = bad_uw_thresh;
    o->M = M; o->Nsym = Nsym; o->Nbits = Nsym * (M == 2 ? 1 : 2); o->bpf = UW_BITS + n;
    memcpy(o->uw, uw, UW_BITS);
    o->row_ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(o->m + 1));
    o->col_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)o->E);
    memcpy(o->row_ptr, row_ptr, sizeof(int32_t) * (size_t)(o->m + 1));
    memcpy(o->col_idx, col_idx, sizeof(int32_t) * (size_t)o->E);
    /* column view: edges of a column in ascending row order */
    o->col_ptr = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    o->col_edge = (int32_t *)malloc(sizeof(int32_t) * (size_t)o->E);
    for (int e = 0; e < o->E; e++) o->col_ptr[col_idx[e] + 1]++;
    for (int v = 0; v < n; v++) o->col_ptr[v + 1] += o->col_ptr[v];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(fill, o->col_ptr, sizeof(int32_t) * (size_t)n);
    for (int r = 0; r < o->m; r++)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) o->col_edge[fill[col_idx[e]]++] = e;
    free(fill);
    /* ln I0 at multiples of 1/8 (power series, double), phi at the centres of 32 bins per octave */
    for (int j = 0; j <= LNI0_N + 1; j++) {
        const double x = j / 8.0;
        double term = 1.0, sum = 1.0;
        for (int t = 1; t < 400; t++) { term *= (x * x / 4.0) / ((double)t * t); sum += term; if (term < sum * 1e-17) break; }
        o->lnI0[j] = (float)log(sum);
    }
    for (int i = 0; i < PHI_N; i++) {
        const int oct = i / PHI_STEPS, st = i % PHI_STEPS;
        const double x0 = ldexp(1.0 + (double)st / PHI_STEPS, PHI_LO_EXP + oct);
        const double xc = ldexp(1.0 + (st + 0.5) / PHI_STEPS, PHI_LO_EXP + oct);
        o->phi[i] = (float)(-log(tanh(xc / 2.0)));
        if (x0 >= (double)PHI_X_HI) o->phi[i] = 0.0f;
        if (x0 <= (double)PHI_X_LO) o->phi[i] = 10.0f;
    }
    o->llr2 = (float *)calloc((size_t)2 * o->bpf, sizeof(float));
    return o;
}

void oracle_ldpc_destroy(LDPC_ORACLE *o)
{
    if (!o) return;
    free(o->row_ptr); free(o->col_idx); free(o->col_ptr); free(o->col_edge); free(o->llr2); free(o);
}

/* Experiment knobs of tools/ldpc_precision.py (which of the product's precision choices moves frames across the decoding edge):
 * bit 0: soft bits stay float32 (no binary16 rounding); bit 1: phi evaluated exactly (libm, double) instead of by table;
 * bit 2: table phi with linear interpolation inside a bin; bits 8..15: table bins per octave when not 32 (evaluated exactly at the
 * bin centre, i.e. what a finer table would hold). 0 = the product's arithmetic, the only setting the parity tests use. */
static int g_ldpc_experiment = 0;
void oracle_ldpc_experiment(int flags) { g_ldpc_experiment = flags; }

/* float -> binary16 -> float, round to nearest even, subnormal halves kept, overflow to infinity (unreachable: |LLR| <= 24) */
float oracle_f16_round(float x)
{
    if (g_ldpc_experiment & 1) return x;
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    float r;
    if (a >= 0x7f800000u) return x;                               /* inf / nan */
    if (a >= 0x477ff000u) { a = 0x7f800000u; }                    /* >= 65520 rounds to infinity */
    else if (a >= 0x38800000u) {                                  /* normal half: keep 10 mantissa bits */
        const uint32_t rem = a & 0x1fffu, base = a & ~0x1fffu;
        a = base + ((rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) ? 0x2000u : 0u);
    } else {                                                      /* subnormal half: multiples of 2^-24 */
        float f;
        memcpy(&f, &a, 4);
        const float q = f * 16777216.0f;                          /* exact */
        const float n = nearbyintf(q);                            /* default rounding mode: nearest even */
        f = n * (1.0f / 16777216.0f);
        memcpy(&a, &f, 4);
    }
    a |= sign;
    memcpy(&r, &a, 4);
    return r;
}

static float ln_i0(const LDPC_ORACLE *o, float x)
{
    if (!(x < 32.0f)) return o->lnI0[LNI0_N] + (x - 32.0f);
    const float xs = x * 8.0f;
    const int j = (int)xs;
    const float f = xs - (float)j;
    const float t0 = o->lnI0[j], t1 = o->lnI0[j + 1];
    return t0 + (f * (t1 - t0));
}

/* [UPSTREAM-RECALLED codec2 mpdecode_core.c: logbesseli0, CML's piecewise-quadratic fit of ln I0] in the float32 operation order the
 * product defines for it (pirip_amd/csrc/fsk_device.hpp: logbesseli0_upstream): coefficients picked by range, then
 * (((c2 x) x) + (c1 x)) + c0, every product and sum rounded once */
static float logbesseli0_upstream(float x)
{
    float c2 = 0.226f, c1 = 0.0125f, c0 = -0.0012f;
    if (x >= 1.0f) { c2 = 0.1245f; c1 = 0.2177f; c0 = -0.108f; }
    if (x >= 2.0f) { c2 = 0.0288f; c1 = 0.6314f; c0 = -0.5645f; }
    if (x >= 5.0f) { c2 = 0.002f; c1 = 0.9048f; c0 = -1.2997f; }
    if (x >= 20.0f) { c2 = 0.0f; c1 = 0.9867f; c0 = -2.2053f; }
    const float q = (c2 * x) * x;
    const float l = c1 * x;
    return (q + l) + c0;
}

static float phi_lookup(const LDPC_ORACLE *o, float x)
{
    if (g_ldpc_experiment & 8) {                           /* bit 3: exact phi WITHOUT the table's range limits (messages saturate at 1e3, like the independent receiver's) */
        double xx = x > 1e-300 ? (double)x : 1e-300;
        const double e = exp(-xx), v = log1p(e) - log1p(-e);
        return (float)(v > 1e3 ? 1e3 : v);
    }
    const float lo = PHI_X_LO;
    if (!(x >= lo)) return 10.0f;
    if (x >= PHI_X_HI) return 0.0f;
    if (g_ldpc_experiment & 2) return (float)(-log(tanh((double)x / 2.0)));
    if (g_ldpc_experiment & 0xff00) {                      /* a table of `steps` bins per octave: the value at the centre of x's bin, or (bit 2) linear between its edges */
        const int steps = (g_ldpc_experiment >> 8) & 0xff;
        int ex; const double mant = frexp((double)x, &ex);   /* x = mant * 2^ex, mant in [0.5, 1) */
        const int st = (int)((mant * 2.0 - 1.0) * steps);
        if (g_ldpc_experiment & 4) {
            const double x0 = ldexp(1.0 + (double)st / steps, ex - 1), x1 = ldexp(1.0 + (double)(st + 1) / steps, ex - 1);
            const float p0 = (float)(-log(tanh(x0 / 2.0))), p1 = (float)(-log(tanh(x1 / 2.0)));
            const float f = (float)(((double)x - x0) / (x1 - x0));
            return p0 + f * (p1 - p0);
        }
        const double xc = ldexp(1.0 + (st + 0.5) / steps, ex - 1);
        return (float)(-log(tanh(xc / 2.0)));
    }
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const int idx = (int)(bits >> 18) - (int)((uint32_t)(127 + PHI_LO_EXP) << 5);
    if (g_ldpc_experiment & 4) {                           /* linear interpolation between bin EDGE values (exact at the edges) */
        const int oct = idx / PHI_STEPS, st = idx % PHI_STEPS;
        const double x0 = ldexp(1.0 + (double)st / PHI_STEPS, PHI_LO_EXP + oct), x1 = ldexp(1.0 + (double)(st + 1) / PHI_STEPS, PHI_LO_EXP + oct);
        const float p0 = (float)(-log(tanh(x0 / 2.0))), p1 = (float)(-log(tanh(x1 / 2.0)));
        if (g_ldpc_experiment & 16) {                      /* bit 4: the pair (p0 - d, d), d = p1 - p0, stored as binary16; value = fma(d, t, p0 - d), t = 1.frac in [1, 2) */
            const int save = g_ldpc_experiment; g_ldpc_experiment &= ~1;
            const float dh = oracle_f16_round(p1 - p0), bh = oracle_f16_round(p0 - (p1 - p0));
            g_ldpc_experiment = save;
            uint32_t tb = ((bits << 5) & 0x007fffe0u) | 0x3f800000u; float t; memcpy(&t, &tb, 4);
            return fmaf(dh, t, bh);
        }
        if (g_ldpc_experiment & 32) {                      /* bit 5: the pair (base, slope per unit x) as binary16: value = fma(slope, x, base), base = p0 - slope x0 with the ROUNDED slope */
            const int save = g_ldpc_experiment; g_ldpc_experiment &= ~1;
            const float sh = oracle_f16_round((float)(((double)p1 - (double)p0) / (x1 - x0)));
            const float bh = oracle_f16_round((float)((double)p0 - (double)sh * x0));
            g_ldpc_experiment = save;
            return fmaf(sh, x, bh);
        }
        const float f = (float)(((double)x - x0) / (x1 - x0));
        return p0 + f * (p1 - p0);
    }
    return o->phi[idx];
}

/* soft decisions of one demodulator call ([m][sym] magnitudes) -> Nbits LLRs (positive = bit 0) */
/* The receiver's DEFINED summation order for a frame's signal / noise terms: that of a 64-lane wave reduction. Lane l first adds
 * its terms l, l + 64, ... in index order; then the lanes combine: + the lane 1, 2, 4, 8 below inside rows of 16 lanes (lanes whose
 * source lies outside the row add +0), rows 1 and 3 + the total of the row below (its lane 15), rows 2 and 3 + lane 31; lane 63
 * holds the sum. codec2's fsk_demod_core adds serially; the two differ in the last bit at most, below the binary16 rounding of the
 * soft bits, and the wave order costs the GPU 14 instructions where the serial one costs 100 dependent adds per frame. */
static float wave_order_sum(const float *x, int n)
{
    float v[64], t[64];
    for (int l = 0; l < 64; l++) { v[l] = 0.0f; for (int i = l; i < n; i += 64) v[l] = v[l] + x[i]; }
    for (int s = 1; s <= 8; s <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = v[l] + ((l & 15) >= s ? v[l - s] : 0.0f);
        memcpy(v, t, sizeof(v));
    }
    for (int l = 0; l < 64; l++) t[l] = v[l] + (((l >> 4) & 1) ? v[(l & ~15) - 1] : 0.0f);      /* rows 1, 3 += lane 15 / 47 */
    memcpy(v, t, sizeof(v));
    for (int l = 0; l < 64; l++) t[l] = v[l] + (l >= 32 ? v[31] : 0.0f);                          /* rows 2, 3 += lane 31 */
    return t[63];
}

void oracle_ldpc_llr(const LDPC_ORACLE *o, const float *r, float *llr)
{
    float *ts = (float *)malloc(sizeof(float) * 2 * (size_t)o->Nsym);
    for (int i = 0; i < o->Nsym; i++) {
        float sum = 0.f, mx = 0.f;
        for (int m = 0; m < o->M; m++) { const float v = r[m * o->Nsym + i]; const float p = v * v; sum = sum + p; mx = p > mx ? p : mx; }
        ts[i] = mx;
        ts[o->Nsym + i] = (sum - mx) / (float)(o->M - 1);
    }
    float sig = wave_order_sum(ts, o->Nsym), nse = wave_order_sum(ts + o->Nsym, o->Nsym);
    free(ts);
    sig = sig / (float)o->Nsym;
    nse = (nse / (float)o->Nsym) + 1e-12f;
    const float a2 = sig - nse;
    const float amp = a2 > 0.f ? sqrtf(a2) : 0.f;
    /* the frame's factor: Rician 2 A / sigma^2; upstream 2 * SNRest / v_est, so that the metric's argument 2 * SNRest * |r| / v_est is g * |r| */
    float g;
    if (o->llr_map == LLR_RICIAN) g = (2.0f * amp) / nse;
    else g = amp > 0.f ? (2.0f * (sig / nse)) / amp : 0.f;
    const float lmax = o->llr_map == LLR_RICIAN ? LLR_MAX : LLR_MAX_UPSTREAM;
    const int bps = o->M == 2 ? 1 : 2;
    for (int i = 0; i < o->Nsym; i++) {
        float L[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < o->M; m++) L[m] = o->llr_map == LLR_RICIAN ? ln_i0(o, g * r[m * o->Nsym + i]) : logbesseli0_upstream(g * r[m * o->Nsym + i]);
        float l0, l1 = 0.f;
        if (o->M == 2) l0 = L[0] - L[1];
        else {
            l0 = (L[0] > L[1] ? L[0] : L[1]) - (L[2] > L[3] ? L[2] : L[3]);
            l1 = (L[0] > L[2] ? L[0] : L[2]) - (L[1] > L[3] ? L[1] : L[3]);
        }
        l0 = l0 > lmax ? lmax : (l0 < -lmax ? -lmax : l0);
        l1 = l1 > lmax ? lmax : (l1 < -lmax ? -lmax : l1);
        llr[bps * i] = oracle_f16_round(l0);
        if (bps == 2) llr[2 * i + 1] = oracle_f16_round(l1);
    }
}

/* flooding sum-product; returns the iterations run, *pcc = satisfied checks of the final hard decisions */
int oracle_ldpc_decode(const LDPC_ORACLE *o, const float *llr_in, uint8_t *hard, int *pcc)
{
    float *Q = (float *)malloc(sizeof(float) * (size_t)o->n), *r = (float *)calloc((size_t)o->E, sizeof(float));
    float *llr = (float *)malloc(sizeof(float) * (size_t)o->n);
    for (int v = 0; v < o->n; v++) llr[v] = oracle_f16_round(llr_in[v]);     /* the decoder's input format is binary16 */
    for (int v = 0; v < o->n; v++) Q[v] = llr[v];
    int iter = 0, ok = 0;
    for (int it = 1; it <= o->max_iter; it++) {
        for (int row = 0; row < o->m; row++) {
            const int e0 = o->row_ptr[row], e1 = o->row_ptr[row + 1];
            float S = 0.0f;
            unsigned sg = 0;
            for (int e = e0; e < e1; e++) {
                const float q = Q[o->col_idx[e]] - r[e];
                sg ^= (q < 0.0f) ? 1u : 0u;
                S = S + phi_lookup(o, fabsf(q));
            }
            float nr[64];
            for (int e = e0; e < e1; e++) {
                const float q = Q[o->col_idx[e]] - r[e];
                const float a = phi_lookup(o, fabsf(q));
                const float mag = phi_lookup(o, S - a);
                const unsigned neg = sg ^ ((q < 0.0f) ? 1u : 0u);
                nr[e - e0] = neg ? -mag : mag;
            }
            for (int e = e0; e < e1; e++) r[e] = nr[e - e0];
        }
        for (int v = 0; v < o->n; v++) {
            float acc = llr[v];
            for (int j = o->col_ptr[v]; j < o->col_ptr[v + 1]; j++) acc = acc + r[o->col_edge[j]];
            Q[v] = acc;
            hard[v] = acc < 0.0f ? 1 : 0;
        }
        ok = 0;
        for (int row = 0; row < o->m; row++) {
            unsigned x = 0;
            for (int e = o->row_ptr[row]; e < o->row_ptr[row + 1]; e++) x ^= hard[o->col_idx[e]];
            ok += !x;
        }
        iter = it;
        if (ok == o->m) break;
    }
    free(Q); free(r); free(llr);
    *pcc = ok;
    return iter;
}

uint16_t oracle_crc16(const uint8_t *bytes, int n)
{
    uint16_t crc = 0xFFFF;
    while (n--) {
        uint8_t x = (uint8_t)(crc >> 8) ^ *bytes++;
        x ^= x >> 4;
        crc = (uint16_t)((crc << 8) ^ ((uint16_t)x << 12) ^ ((uint16_t)x << 5) ^ (uint16_t)x);
    }
    return crc;
}

/* one demodulator call of the receiver: rx_filt -> status byte, k/8 payload bytes (zeros when none), info[10] */
int oracle_ldpc_rx_call(LDPC_ORACLE *o, const float *rx_filt, uint8_t *payload, int32_t *info)
{
    const int bpf = o->bpf, Nbits = o->Nbits;
    memmove(o->llr2, o->llr2 + Nbits, sizeof(float) * (size_t)(2 * bpf - Nbits));
    if (rx_filt) oracle_ldpc_llr(o, rx_filt, o->llr2 + 2 * bpf - Nbits);
    else for (int b = 0; b < Nbits; b++) o->llr2[2 * bpf - Nbits + b] = 0.0f;
    int next = o->state;
#define UWERR(p, out) do { int e_ = 0; for (int u = 0; u < UW_BITS; u++) e_ += ((o->llr2[(p) + u] < 0.0f) ? 1 : 0) ^ o->uw[u]; (out) = e_; } while (0)
    if (o->state == 0) {
        int best = 255, bi = 0;
        for (int i = 0; i < bpf; i++) { int e; UWERR(i, e); if (e < best) { best = e; bi = i; } }
        o->uw_err = best;
        if (best <= o->uw_thresh1) { next = 1; o->loc = bi; o->bad_uw = 0; }
    } else {
        o->loc -= Nbits;
        if (o->loc < 0) {
            o->loc += bpf;
            UWERR(o->loc, o->uw_err);
            if (o->uw_err > o->uw_thresh2) { o->bad_uw++; if (o->bad_uw >= o->bad_uw_thresh) next = 0; }
            else o->bad_uw = 0;
        }
    }
    int status = 0, iter = 0, pcc = 0, crc_ok = 0, pos = -1, eraw = 0;
    memset(payload, 0, (size_t)o->k / 8);
    if (next == 1) {
        status |= RX_SYNC;
        if (o->loc >= 0 && o->loc < Nbits) {
            pos = o->loc;
            uint8_t *hard = (uint8_t *)malloc((size_t)o->n);
            iter = oracle_ldpc_decode(o, o->llr2 + o->loc + UW_BITS, hard, &pcc);
            for (int v = 0; v < o->n; v++) eraw += ((o->llr2[o->loc + UW_BITS + v] < 0.0f) ? 1 : 0) != (hard[v] != 0);
            const int nbytes = o->k / 8;
            for (int b = 0; b < nbytes; b++) {
                unsigned byte = 0;
                for (int i = 0; i < 8; i++) byte |= (unsigned)hard[8 * b + i] << (7 - i);
                payload[b] = (uint8_t)byte;
            }
            crc_ok = oracle_crc16(payload, nbytes - 2) == (uint16_t)((payload[nbytes - 2] << 8) | payload[nbytes - 1]);
            if (crc_ok) status |= RX_BITS;
            if (pcc != o->m) status |= RX_BIT_ERRORS;
            free(hard);
        }
    }
    o->state = next;
    info[0] = o->state; info[1] = o->loc; info[2] = o->uw_err; info[3] = o->bad_uw; info[4] = iter; info[5] = pcc; info[6] = pos; info[7] = crc_ok; info[8] = eraw; info[9] = 0;
    return status;
}
