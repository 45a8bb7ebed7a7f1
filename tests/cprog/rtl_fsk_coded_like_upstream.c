/* tests/cprog/rtl_fsk_coded_like_upstream.c -- the coded receive loop of upstream's rtl_fsk.c [UPSTREAM-RECALLED librtlsdr src/rtl_fsk.c,
 * --code mode], written against libcodec2's FreeDV API names only: freedv_open_advanced(FREEDV_MODE_FSK_LDPC, &adv), freedv_get_fsk +
 * fsk_set_freq_est_limits, then
 *     while (fread(freedv_nin() samples)) { nbytes = freedv_rawdatacomprx(); status = freedv_get_rx_status(); fwrite(status, bytes) }
 * i.e. the `-b` record stream script/frame_repeater:36 pipes into tx/frame_repeater.c:55-62. It includes only include/pirip_hip.h and
 * links libpirip_hip.so: the coded half of the library-level boundary (/root/reference/build_rtlsdr.sh:9 links rtl_fsk
 * against libcodec2). Plain C. Also walks the Tx-side helpers /root/reference/tx/rpitx_fsk.cpp:33-40,75-83 declares by hand.
 *
 *   rtl_fsk_coded_like_upstream CODE M Fs Rs fsk_lower fsk_upper [v] < u8 IQ > records     (u8 -> float as csdr convert_u8_f) */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pirip_hip.h"

int main(int argc, char **argv)
{
    if (argc < 7) return 1;
    struct freedv_advanced adv;
    memset(&adv, 0x5a, sizeof(adv));                /* rpitx_fsk leaves first_tone / tone_spacing uninitialised: so do we */
    adv.codename = argv[1]; adv.M = atoi(argv[2]); adv.Fs = atoi(argv[3]); adv.Rs = atoi(argv[4]);
    struct freedv *freedv = freedv_open_advanced(FREEDV_MODE_FSK_LDPC, &adv);
    if (freedv == NULL) return 3;
    struct FSK *fsk = freedv_get_fsk(freedv);
    fsk_set_freq_est_limits(fsk, atoi(argv[5]), atoi(argv[6]));
    if (argc > 7) { freedv_set_verbose(freedv, 2); freedv_set_test_frames(freedv, 1); }
    const int bits_per_modem_frame = freedv_get_bits_per_modem_frame(freedv);
    const int bytes_per_modem_frame = bits_per_modem_frame / 8;
    assert(bits_per_modem_frame % 8 == 0 && bytes_per_modem_frame > 2);

    /* Tx side, as rpitx_fsk.cpp:75-83,394-395 builds a frame: CRC16 into the last 16 data bits, UW + data + parity */
    {
        const int bits_per_frame = freedv_tx_fsk_ldpc_bits_per_frame(freedv);
        uint8_t *data_bits = (uint8_t *)malloc((size_t)bits_per_modem_frame), *frame = (uint8_t *)malloc((size_t)bits_per_frame);
        unsigned char *bytes = (unsigned char *)malloc((size_t)bytes_per_modem_frame);
        ofdm_generate_payload_data_bits(data_bits, bits_per_modem_frame);
        freedv_pack(bytes, data_bits, bits_per_modem_frame - 16);
        unsigned short crc16 = freedv_gen_crc16(bytes, bytes_per_modem_frame - 2);
        unsigned char crc_bytes[2] = {(unsigned char)(crc16 >> 8), (unsigned char)(crc16 & 0xff)};
        freedv_unpack(data_bits + bits_per_modem_frame - 16, crc_bytes, 16);
        freedv_tx_fsk_ldpc_framer(freedv, frame, data_bits);
        assert(bits_per_frame > bits_per_modem_frame + 32 && memcmp(frame + 32, data_bits, (size_t)bits_per_modem_frame) == 0);
        unsigned char check[9] = {'1', '2', '3', '4', '5', '6', '7', '8', '9'};
        assert(freedv_gen_crc16(check, 9) == 0x29B1);          /* CRC-16/CCITT-FALSE check value */
        fprintf(stderr, "tx: bits_per_frame %d data_bits_per_frame %d crc16 %04x\n", bits_per_frame, bits_per_modem_frame, crc16);
        free(data_bits); free(frame); free(bytes);
    }

    const int nmax = freedv_get_n_max_modem_samples(freedv);
    unsigned char *raw = (unsigned char *)malloc(2 * (size_t)nmax);
    COMP *rx = (COMP *)malloc(sizeof(COMP) * (size_t)nmax);
    unsigned char *bytes_out = (unsigned char *)malloc((size_t)bytes_per_modem_frame);
    long calls = 0, frames = 0;
    int nin = freedv_nin(freedv);
    assert(nin > 0 && nin <= nmax);
    while (fread(raw, 2, (size_t)nin, stdin) == (size_t)nin) {
        for (int i = 0; i < nin; i++) {                        /* csdr convert_u8_f: x / (UCHAR_MAX / 2.0) - 1.0 */
            rx[i].real = (float)((double)raw[2 * i] / 127.5 - 1.0);
            rx[i].imag = (float)((double)raw[2 * i + 1] / 127.5 - 1.0);
        }
        int nbytes = freedv_rawdatacomprx(freedv, bytes_out, rx);
        unsigned char rx_status = (unsigned char)freedv_get_rx_status(freedv);
        assert(nbytes == 0 || nbytes == bytes_per_modem_frame);
        assert((nbytes != 0) == ((rx_status & FREEDV_RX_BITS) != 0));
        if (nbytes == 0) memset(bytes_out, 0, (size_t)bytes_per_modem_frame);
        fwrite(&rx_status, 1, 1, stdout);                      /* -b: one status byte + the data bytes, zeros when no frame */
        fwrite(bytes_out, 1, (size_t)bytes_per_modem_frame, stdout);
        if (nbytes) frames++;
        {   /* what a status display reads */
            int sync = -1; float snr = -1.0f;
            struct MODEM_STATS ext;
            freedv_get_modem_stats(freedv, &sync, &snr);
            freedv_get_modem_extended_stats(freedv, &ext);
            assert(sync == ((rx_status & FREEDV_RX_SYNC) != 0) && ext.sync == sync && ext.Nc == adv.M && snr == ext.snr_est);
        }
        calls++;
        nin = freedv_nin(freedv);
        assert(nin > 0 && nin <= nmax);
    }
    fprintf(stderr, "calls %ld frames %ld f_est %.1f %.1f SNRest %.3f\n", calls, frames, fsk->f_est[0], fsk->f_est[adv.M - 1], fsk->SNRest);
    freedv_close(freedv);
    free(raw); free(rx); free(bytes_out);
    return 0;
}
