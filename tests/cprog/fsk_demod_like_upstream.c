/* tests/cprog/fsk_demod_like_upstream.c -- a receive loop written the way codec2's own programs drive libcodec2
 * (fsk_demod.c / rtl_fsk.c [UPSTREAM-RECALLED]): fsk_create_hbr, fsk_set_freq_est_limits, then
 *     while (fread(nin samples)) { fsk_demod(); fwrite(fsk->Nbits bits); ... fsk->f_est[], fsk->norm_rx_timing ... }
 * with DIRECT reads of struct FSK fields and a full-layout MODEM_STATS, plus libcsdr's firdes_lowpass_f(..., window_t).
 * It includes only include/pirip_hip.h and links libpirip_hip.so: the library-level boundary test of SURVEY.md 8b
 * (what /root/reference/build_rtlsdr.sh:9 links rtl_fsk against). Plain C, no HIP, no C++.
 *
 *   fsk_demod_like_upstream M Fs Rs P fsk_lower fsk_upper [eye] < complex_s16 > bits ; per-frame text goes to stderr
 * A seventh argument makes it a program that plots the eye diagram (fsk_stats_normalise_eye before the first frame: the shim's
 * opt-in for MODEM_STATS.rx_eye); without it the handle stays on its specialised kernel and neyetr reads 0. */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include "pirip_hip.h"

int main(int argc, char **argv)
{
    if (argc < 7) return 1;
    const int M = atoi(argv[1]), Fs = atoi(argv[2]), Rs = atoi(argv[3]), P = atoi(argv[4]);
    struct FSK *fsk = fsk_create_hbr(Fs, Rs, M, P, 50, 1200, 1200);
    fsk_set_freq_est_limits(fsk, atoi(argv[5]), atoi(argv[6]));
    if (argc > 7) fsk_stats_normalise_eye(fsk, 1);
    assert(fsk->Nbits == 50 * (M == 2 ? 1 : 2) && fsk->N == fsk->Ts * fsk->Nsym && fsk->Ndft > 0 && fsk->Sf != NULL);
    uint8_t *bitbuf = (uint8_t *)malloc((size_t)fsk->Nbits);
    COMP *modbuf = (COMP *)malloc(sizeof(COMP) * (size_t)(fsk->N + fsk->Ts * 2));
    int16_t *rawbuf = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)(fsk->N + fsk->Ts * 2));
    struct MODEM_STATS stats;
    long frame = 0;
    while (fread(rawbuf, sizeof(int16_t) * 2, fsk_nin(fsk), stdin) == fsk_nin(fsk)) {
        for (unsigned i = 0; i < fsk_nin(fsk); i++) {
            modbuf[i].real = ((float)rawbuf[2 * i]) / PIRIP_FDMDV_SCALE;
            modbuf[i].imag = ((float)rawbuf[2 * i + 1]) / PIRIP_FDMDV_SCALE;
        }
        fsk_demod(fsk, bitbuf, modbuf);
        fwrite(bitbuf, 1, (size_t)fsk->Nbits, stdout);
        fsk_get_demod_stats(fsk, &stats);
        /* the eye diagram as fsk_demod.c's -t mode walks it */
        float eyesum = 0, eyemax = 0;
        for (int i = 0; i < stats.neyetr; i++)
            for (int j = 0; j < stats.neyesamp; j++) { eyesum += stats.rx_eye[i][j]; if (stats.rx_eye[i][j] > eyemax) eyemax = stats.rx_eye[i][j]; }
        float sfmax = 0; int sfi = 0;
        for (int i = 0; i < fsk->Ndft; i++) if (fsk->Sf[i] > sfmax) { sfmax = fsk->Sf[i]; sfi = i; }
        fprintf(stderr, "%ld nin %d f_est %.3f %.3f timing %.6f SNRest %.5e ppm %.4f EbNodB %.4f snr_est %.4f clock %.4f rx_timing %.5f sfpeak %d neyetr %d neyesamp %d eyesum %.6f eyemax %.6f\n",
                frame, fsk->nin, fsk->f_est[0], fsk->f_est[M - 1], fsk->norm_rx_timing, fsk->SNRest, fsk->ppm, fsk->EbNodB,
                stats.snr_est, stats.clock_offset, stats.rx_timing, sfi, stats.neyetr, stats.neyesamp, eyesum, eyemax);
        frame++;
    }
    /* libcsdr: the window is an argument */
    float taps[79], taps_h[79];
    assert(firdes_filter_len(0.05f) == 79);
    firdes_lowpass_f(taps, 79, 0.5f / 45, WINDOW_HAMMING);
    firdes_lowpass_f_hamming(taps_h, 79, 0.5f / 45);
    for (int i = 0; i < 79; i++) assert(taps[i] == taps_h[i]);
    firdes_lowpass_f(taps, 79, 0.5f / 45, WINDOW_BOXCAR);
    float s = 0; for (int i = 0; i < 79; i++) s += taps[i];
    assert(s > 0.999f && s < 1.001f && taps[0] != taps_h[0]);
    fsk_destroy(fsk);
    free(bitbuf); free(modbuf); free(rawbuf);
    return 0;
}
