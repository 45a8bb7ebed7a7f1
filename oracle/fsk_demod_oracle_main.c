/* oracle/fsk_demod_oracle_main.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; PARITY UNPINNED).
 * Command-line front end restating codec2 src/fsk_demod.c main() [UPSTREAM-RECALLED];
 * argv forms pinned by /root/reference/README.md:105,109 and
 * /root/reference/test/loopback_rtl_sdr.sh:16:
 *   fsk_demod [--fsk_lower Hz] [--fsk_upper Hz] [-d|-c] [-p P] [--mask spacing] [-s] M Fs Rs in out
 * Used by tests to compare the product's fsk_demod stdout byte-for-byte. */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fsk_oracle.h"

int main(int argc, char **argv)
{
    int complex_in = 0, u8_in = 0, soft = 0, P = ORACLE_FSK_DEFAULT_P, mask = 0;
    int user_lower = 0, user_upper = 0, fsk_lower = 0, fsk_upper = 0, nsym = ORACLE_FSK_DEFAULT_NSYM;
    static struct option lopts[] = {
        {"fsk_lower", required_argument, 0, 'b'}, {"fsk_upper", required_argument, 0, 'u'},
        {"mask", required_argument, 0, 'm'}, {"cu8", no_argument, 0, 'd'}, {"cs16", no_argument, 0, 'c'},
        {"conv", required_argument, 0, 'p'}, {"soft-dec", no_argument, 0, 's'},
        {"nsym", required_argument, 0, 'n'}, {"testmode", no_argument, 0, 't'}, {0, 0, 0, 0}};
    int o, oi;
    while ((o = getopt_long(argc, argv, "dcsp:b:u:m:n:t::", lopts, &oi)) != -1) {
        switch (o) {
        case 'd': u8_in = 1; complex_in = 1; break;
        case 'c': complex_in = 1; break;
        case 's': soft = 1; break;
        case 'p': P = atoi(optarg); break;
        case 'b': fsk_lower = atoi(optarg); user_lower = 1; break;
        case 'u': fsk_upper = atoi(optarg); user_upper = 1; break;
        case 'm': mask = atoi(optarg); break;
        case 'n': nsym = atoi(optarg); break;
        case 't': break;
        default: return 2;
        }
    }
    if (argc - optind < 5) { fprintf(stderr, "usage: %s [opts] M Fs Rs in out\n", argv[0]); return 2; }
    int M = atoi(argv[optind]), Fs = atoi(argv[optind + 1]), Rs = atoi(argv[optind + 2]);
    FILE *fin = strcmp(argv[optind + 3], "-") ? fopen(argv[optind + 3], "rb") : stdin;
    FILE *fout = strcmp(argv[optind + 4], "-") ? fopen(argv[optind + 4], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "couldn't open files\n"); return 1; }

    struct fsk_oracle_recalled rc;                      /* the recalled constants; PIRIP_RECALLED="field=value,..." flips them (pin_against_ref.py) */
    oracle_fsk_recalled_defaults(&rc);
    if (oracle_fsk_recalled_from_env(&rc) < 0) { fprintf(stderr, "PIRIP_RECALLED: unknown field\n"); return 2; }
    struct ORACLE_FSK *fsk = oracle_fsk_create_recalled(Fs, Rs, M, P, nsym, ORACLE_FSK_NONE, mask ? mask : 100, &rc);
    if (!user_lower) fsk_lower = complex_in ? -Fs / 2 : 0;
    if (!user_upper) fsk_upper = Fs / 2;
    fprintf(stderr, "Setting estimator limits to %d to %d Hz.\n", fsk_lower, fsk_upper);
    oracle_fsk_set_freq_est_limits(fsk, fsk_lower, fsk_upper);
    if (mask) oracle_fsk_set_freq_est_alg(fsk, 1);

    int maxnin = fsk->N + fsk->Ts * 2;
    COMP *modbuf = malloc(sizeof(COMP) * maxnin);
    uint8_t *bitbuf = malloc(fsk->Nbits);
    float *sd = malloc(sizeof(float) * M * fsk->Nsym);
    void *raw = malloc(4 * sizeof(int16_t) * maxnin);
    size_t bps = u8_in ? 2 : (complex_in ? 4 : 2);
    while (fread(raw, bps, oracle_fsk_nin(fsk), fin) == oracle_fsk_nin(fsk)) {
        int nin = oracle_fsk_nin(fsk);
        for (int i = 0; i < nin; i++) {
            if (u8_in) {
                modbuf[i].real = ((float)((uint8_t *)raw)[2 * i] - (double)rc.u8d_offset) / (double)rc.u8d_scale;      /* (x - 127.0) / 128.0 as recalled */
                modbuf[i].imag = ((float)((uint8_t *)raw)[2 * i + 1] - (double)rc.u8d_offset) / (double)rc.u8d_scale;
            } else if (complex_in) {
                modbuf[i].real = ((float)((int16_t *)raw)[2 * i]) / rc.s16_scale;                 /* / FDMDV_SCALE */
                modbuf[i].imag = ((float)((int16_t *)raw)[2 * i + 1]) / rc.s16_scale;
            } else {
                modbuf[i].real = ((float)((int16_t *)raw)[i]) / rc.s16_scale;
                modbuf[i].imag = 0.0;
            }
        }
        memset(bitbuf, 0, fsk->Nbits); memset(sd, 0, sizeof(float) * M * fsk->Nsym);
        oracle_fsk_demod_core(fsk, bitbuf, sd, modbuf);
        if (soft) fwrite(sd, sizeof(float), M * fsk->Nsym, fout);
        else fwrite(bitbuf, 1, fsk->Nbits, fout);
        if (fout == stdout) fflush(fout);
    }
    oracle_fsk_destroy(fsk);
    return 0;
}
