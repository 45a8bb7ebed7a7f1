/* oracle/ref_csdr_main.c -- TEST INFRASTRUCTURE ONLY: a driver around UPSTREAM's libcsdr.c functions, linked by
 * oracle/build_ref.sh against the csdr checkout (never against this repo's restatement). It replays the three-process pipe of
 * /root/reference/README.md:109 in one process:
 *     csdr_path <decimation> [transition_bw] < u8 IQ > s16 IQ        (convert_u8_f | fir_decimate_cc D | convert_f_s16)
 *     csdr_path -f <decimation> [transition_bw] < u8 IQ > f32 IQ     (without the s16 hop: what rtl_fsk does in-process)
 * calling convert_u8_f, firdes_filter_len, firdes_lowpass_f, fir_decimate_cc and convert_f_s16 by upstream's names and
 * signatures [UPSTREAM-RECALLED libcsdr.h; a mismatch is a compile error, which is the point]. The whole input is presented
 * as ONE buffer (csdr's block size only decides where the dropped tail lies). If libcsdr.c does not compile alone in a given
 * checkout (FFTW-dependent functions), build it with -DUSE_FFTW=0 or add its fft_fftw.c: do not edit upstream's file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libcsdr.h"

int main(int argc, char **argv)
{
    int a = 1, f32_out = 0;
    if (a < argc && !strcmp(argv[a], "-f")) { f32_out = 1; a++; }
    if (a >= argc) { fprintf(stderr, "usage: csdr_path [-f] decimation [transition_bw]\n"); return 2; }
    const int D = atoi(argv[a++]);
    const float tbw = a < argc ? (float)atof(argv[a]) : 0.05f;
    size_t cap = 1 << 20, n = 0;
    unsigned char *in = malloc(cap);
    for (;;) { size_t r = fread(in + n, 1, cap - n, stdin); n += r; if (r == 0) break; if (n == cap) { cap *= 2; in = realloc(in, cap); } }
    const int nsamp = (int)(n / 2);
    float *x = malloc(sizeof(float) * 2 * (size_t)nsamp);
    convert_u8_f(in, x, 2 * nsamp);
    int taps_length = firdes_filter_len(tbw);
    int padded = taps_length + (4 - taps_length % 4) % 4;            /* csdr pads the taps to a multiple of 4 with zeros */
    float *taps = calloc((size_t)padded, sizeof(float));
    firdes_lowpass_f(taps, taps_length, 0.5f / (float)D, WINDOW_DEFAULT);
    complexf *y = malloc(sizeof(complexf) * ((size_t)nsamp / D + 1));
    const int nout = fir_decimate_cc((complexf *)x, y, nsamp, D, taps, padded);
    if (f32_out) fwrite(y, sizeof(complexf), (size_t)nout, stdout);
    else {
        short *s = malloc(sizeof(short) * 2 * (size_t)nout);
        convert_f_s16((float *)y, s, 2 * nout);
        fwrite(s, sizeof(short), 2 * (size_t)nout, stdout);
    }
    return 0;
}
