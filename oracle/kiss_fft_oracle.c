/* oracle/kiss_fft_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned).
 *
 * Restatement of the float32 complex forward FFT codec2 uses inside its FSK
 * frequency estimator [UPSTREAM-RECALLED: codec2 src/kiss_fft.c, src/_kiss_fft_guts.h;
 * the dependency is cloned un-pinned by /root/reference/build_codec2.sh:3-5 and is not
 * present under /root/reference -- see SURVEY.md section 0].
 *
 * What is restated (and why the order matters): the HIP path promises a bit-exact
 * smoothed spectrum Sf, so this file keeps the published algorithm's exact dataflow:
 *   - factorisation: all factors of 4 first, then 2, then odd primes (kf_factor);
 *   - decimation-in-time recursion, leaves copied with stride fstride (kf_work);
 *   - radix-4 and radix-2 butterflies with the same sequence of float adds/multiplies
 *     (kf_bfly4 / kf_bfly2), twiddle k of a stage taken at index k*fstride;
 *   - twiddles = (float)cos/sin of a double phase -2*pi*i/nfft.
 * Only radices 4 and 2 are implemented: every Ndft codec2's fsk_create_core derives is a
 * power of two (Ndft = 2^ceil(log2(Fs/(0.1*Rs)))).
 */
#include <stdlib.h>
#include <math.h>
#include <assert.h>
#include "kiss_fft_oracle.h"

struct kiss_fft_oracle_state {
    int nfft;
    int inverse;
    int factors[2 * 32];
    kiss_fft_oracle_cpx *twiddles;
};

#define C_MUL(m, a, b) do { (m).r = (a).r * (b).r - (a).i * (b).i; \
                            (m).i = (a).r * (b).i + (a).i * (b).r; } while (0)
#define C_ADD(res, a, b) do { (res).r = (a).r + (b).r; (res).i = (a).i + (b).i; } while (0)
#define C_SUB(res, a, b) do { (res).r = (a).r - (b).r; (res).i = (a).i - (b).i; } while (0)
#define C_ADDTO(res, a) do { (res).r += (a).r; (res).i += (a).i; } while (0)

/* radix-2 butterfly [UPSTREAM-RECALLED kiss_fft.c: kf_bfly2] */
static void bfly2(kiss_fft_oracle_cpx *Fout, size_t fstride, const kiss_fft_oracle_cfg st, int m)
{
    kiss_fft_oracle_cpx *Fout2 = Fout + m;
    const kiss_fft_oracle_cpx *tw1 = st->twiddles;
    kiss_fft_oracle_cpx t;
    do {
        C_MUL(t, *Fout2, *tw1);
        tw1 += fstride;
        C_SUB(*Fout2, *Fout, t);
        C_ADDTO(*Fout, t);
        ++Fout2;
        ++Fout;
    } while (--m);
}

/* radix-4 butterfly, forward transform [UPSTREAM-RECALLED kiss_fft.c: kf_bfly4] */
static void bfly4(kiss_fft_oracle_cpx *Fout, size_t fstride, const kiss_fft_oracle_cfg st, size_t m)
{
    const kiss_fft_oracle_cpx *tw1, *tw2, *tw3;
    kiss_fft_oracle_cpx scratch[6];
    size_t k = m;
    const size_t m2 = 2 * m, m3 = 3 * m;
    tw3 = tw2 = tw1 = st->twiddles;
    do {
        C_MUL(scratch[0], Fout[m], *tw1);
        C_MUL(scratch[1], Fout[m2], *tw2);
        C_MUL(scratch[2], Fout[m3], *tw3);

        C_SUB(scratch[5], *Fout, scratch[1]);
        C_ADDTO(*Fout, scratch[1]);
        C_ADD(scratch[3], scratch[0], scratch[2]);
        C_SUB(scratch[4], scratch[0], scratch[2]);
        C_SUB(Fout[m2], *Fout, scratch[3]);
        tw1 += fstride;
        tw2 += fstride * 2;
        tw3 += fstride * 3;
        C_ADDTO(*Fout, scratch[3]);

        if (st->inverse) {
            Fout[m].r = scratch[5].r - scratch[4].i;
            Fout[m].i = scratch[5].i + scratch[4].r;
            Fout[m3].r = scratch[5].r + scratch[4].i;
            Fout[m3].i = scratch[5].i - scratch[4].r;
        } else {
            Fout[m].r = scratch[5].r + scratch[4].i;
            Fout[m].i = scratch[5].i - scratch[4].r;
            Fout[m3].r = scratch[5].r - scratch[4].i;
            Fout[m3].i = scratch[5].i + scratch[4].r;
        }
        ++Fout;
    } while (--k);
}

/* recursive decimation-in-time driver [UPSTREAM-RECALLED kiss_fft.c: kf_work] */
static void work(kiss_fft_oracle_cpx *Fout, const kiss_fft_oracle_cpx *f, size_t fstride,
                 const int *factors, const kiss_fft_oracle_cfg st)
{
    kiss_fft_oracle_cpx *Fout_beg = Fout;
    const int p = *factors++;   /* the radix */
    const int m = *factors++;   /* stage's fft length / p */
    const kiss_fft_oracle_cpx *Fout_end = Fout + p * m;

    if (m == 1) {
        do {
            *Fout = *f;
            f += fstride;
        } while (++Fout != Fout_end);
    } else {
        do {
            work(Fout, f, fstride * p, factors, st);
            f += fstride;
        } while ((Fout += m) != Fout_end);
    }
    Fout = Fout_beg;
    switch (p) {
    case 2: bfly2(Fout, fstride, st, m); break;
    case 4: bfly4(Fout, fstride, st, (size_t)m); break;
    default: assert(!"kiss_fft_oracle: only radix 2/4 (power-of-two nfft)");
    }
}

/* [UPSTREAM-RECALLED kiss_fft.c: kf_factor] 4s first, then 2s */
static void factor(int n, int *facbuf)
{
    int p = 4;
    double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        *facbuf++ = p;
        *facbuf++ = n;
    } while (n > 1);
}

kiss_fft_oracle_cfg kiss_fft_oracle_alloc(int nfft, int inverse_fft)
{
    kiss_fft_oracle_cfg st = (kiss_fft_oracle_cfg)calloc(1, sizeof(*st));
    assert(st);
    assert(nfft > 1 && (nfft & (nfft - 1)) == 0);
    st->nfft = nfft;
    st->inverse = inverse_fft;
    st->twiddles = (kiss_fft_oracle_cpx *)malloc(sizeof(kiss_fft_oracle_cpx) * (size_t)nfft);
    assert(st->twiddles);
    for (int i = 0; i < nfft; ++i) {
        const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
        double phase = -2 * pi * i / nfft;
        if (st->inverse) phase *= -1;
        st->twiddles[i].r = (float)cos(phase);
        st->twiddles[i].i = (float)sin(phase);
    }
    factor(nfft, st->factors);
    return st;
}

void kiss_fft_oracle_free(kiss_fft_oracle_cfg st)
{
    if (st) { free(st->twiddles); free(st); }
}

void kiss_fft_oracle(kiss_fft_oracle_cfg st, const kiss_fft_oracle_cpx *fin, kiss_fft_oracle_cpx *fout)
{
    assert(fin != fout);
    work(fout, fin, 1, st->factors, st);
}

const kiss_fft_oracle_cpx *kiss_fft_oracle_twiddles(kiss_fft_oracle_cfg st) { return st->twiddles; }
