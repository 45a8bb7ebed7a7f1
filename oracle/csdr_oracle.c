/* oracle/csdr_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; PARITY UNPINNED).
 * See csdr_oracle.h for provenance. Evaluation order = upstream scalar C path
 * (float accumulators, taps applied in ascending index order, no FMA). */
#include <assert.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "csdr_oracle.h"

#define PI ((float)3.14159265358979323846)

/* [UPSTREAM-RECALLED libcsdr.c: convert_u8_f] */
void oracle_convert_u8_f(const unsigned char *input, float *output, int length)
{
    for (int i = 0; i < length; i++) output[i] = ((float)input[i]) / (UCHAR_MAX / 2.0) - 1.0;
}

/* [UPSTREAM-RECALLED libcsdr.c: convert_f_s16] float -> short by C conversion (truncation) */
void oracle_convert_f_s16(const float *input, short *output, int length)
{
    for (int i = 0; i < length; i++) output[i] = input[i] * SHRT_MAX;
}

/* [UPSTREAM-RECALLED libcsdr.c: firdes_filter_len] */
int oracle_firdes_filter_len(float transition_bw)
{
    int result = 4.0 / transition_bw;
    if (result % 2 == 0) result++;
    return result;
}

/* [UPSTREAM-RECALLED libcsdr.c: firdes_wkernel_hamming] */
static float wkernel_hamming(float rate)
{
    rate = 0.5 + rate / 2;
    return 0.54 - 0.46 * cos(2 * PI * rate);
}

/* [UPSTREAM-RECALLED libcsdr.c: firdes_lowpass_f] windowed sinc, normalised to unit DC gain */
void oracle_firdes_lowpass_f_hamming(float *output, int length, float cutoff_rate)
{
    int middle = length / 2;
    output[middle] = 2 * PI * cutoff_rate * wkernel_hamming(0);
    for (int i = 1; i <= middle; i++)
        output[middle - i] = output[middle + i] =
            (sin(2 * PI * cutoff_rate * i) / i) * wkernel_hamming((float)i / middle);
    float sum = 0;
    for (int i = 0; i < length; i++) sum += output[i];
    for (int i = 0; i < length; i++) output[i] /= sum;
}

/* [UPSTREAM-RECALLED libcsdr.c: fir_decimate_cc] direct form, real taps, no zero pre-history */
int oracle_fir_decimate_cc(const float *input, float *output, int input_size, int decimation,
                           const float *taps, int taps_length)
{
    int oi = 0;
    for (int i = 0; i < input_size; i += decimation) {
        if (i + taps_length > input_size) break;
        float acci = 0;
        float accq = 0;
        for (int ti = 0; ti < taps_length; ti++) {
            acci += input[2 * (i + ti)] * taps[ti];
            accq += input[2 * (i + ti) + 1] * taps[ti];
        }
        output[2 * oi] = acci;
        output[2 * oi + 1] = accq;
        oi++;
    }
    return oi;
}

/* [UPSTREAM-RECALLED csdr.c: "fir_decimate_cc" command loop] */
long oracle_csdr_fir_decimate_stream(const float *in_c, long nsamp, float *out_c, long max_out,
                                     int decimation, float transition_bw, int bufsize)
{
    int taps_length = oracle_firdes_filter_len(transition_bw);
    while (bufsize < taps_length * 2) bufsize *= 2;
    int padded = taps_length + 3 - ((taps_length + 3) % 4);
    float *taps = (float *)calloc((size_t)padded, sizeof(float)); assert(taps);
    oracle_firdes_lowpass_f_hamming(taps, taps_length, 0.5 / (float)decimation);
    float *ibuf = (float *)malloc(sizeof(float) * 2 * (size_t)bufsize);
    float *obuf = (float *)malloc(sizeof(float) * 2 * (size_t)bufsize);
    assert(ibuf && obuf);
    long rd = 0, wr = 0;
    if (nsamp < bufsize) goto done;                  /* first fread short -> nothing is emitted */
    memcpy(ibuf, in_c, sizeof(float) * 2 * (size_t)bufsize); rd = bufsize;
    for (;;) {
        int osz = oracle_fir_decimate_cc(ibuf, obuf, bufsize, decimation, taps, padded);
        if (wr + osz > max_out) osz = (int)(max_out - wr);
        memcpy(out_c + 2 * wr, obuf, sizeof(float) * 2 * (size_t)osz); wr += osz;
        int skip = decimation * osz;
        memmove(ibuf, ibuf + 2 * skip, sizeof(float) * 2 * (size_t)(bufsize - skip));
        if (rd + skip > nsamp || wr >= max_out) break;   /* short read ends the stream */
        memcpy(ibuf + 2 * (bufsize - skip), in_c + 2 * rd, sizeof(float) * 2 * (size_t)skip);
        rd += skip;
    }
done:
    free(taps); free(ibuf); free(obuf);
    return wr;
}
