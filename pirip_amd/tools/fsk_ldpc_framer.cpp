// fsk_ldpc_framer -- the framing half of pirip's transmitter, without the RF: what `rpitx_fsk --code NAME` does to its
// input before the bits reach the FSK modulator (/root/reference/tx/rpitx_fsk.cpp), as a CPU tool whose output feeds
// `fsk_mod` (the reference's bench chain does the same with freedv_data_raw_tx, README.md:228,232,257). SURVEY.md 8f-4.
//
//   fsk_ldpc_framer --code FILE [-m M] [--packed] [--source BYTE] [--seq] [--gap BITS] In|- OutOneBitPerByte|-
//       stdin protocol of rpitx_fsk in FSK_LDPC mode (rpitx_fsk.cpp:427-509): records of one burst-control byte followed by
//       data_bits_per_frame bits (one per byte) or, with --packed, data_bits_per_frame/8 bytes. burst_control 1 = first
//       frame of a burst (preamble first, :453-464), 0 = next frame, 2 = end of burst (dummy data frame, nothing sent,
//       :475-494). The last 16 data bits are replaced by the CRC16 of the packed frame (:75-83,467-469). This is what
//       tx/frame_repeater.c:92-104 writes.
//   fsk_ldpc_framer --code FILE --testframes N [--bursts B] [...] /dev/zero Out|-
//       rpitx_fsk's test-frame mode (:366-421): N frames per burst, optional source byte in byte 0 (:382-386) and sequence
//       number (f+1)&0xff in byte 1 (:388-393).
// Output: one bit per byte: preamble | UW data parity | ... per burst, and --gap zero bits between bursts (the carrier is
// off there on the air; a test harness replaces them by silence at the sample level). Bit->symbol order is the
// modulator's: MSB first (rpitx_fsk.cpp:129-141).
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fsk_ldpc.hpp"

using namespace pirip;

int main(int argc, char **argv)
{
    std::string code_path;
    int M = 2, packed = 0, testframes = 0, bursts = 1, source = -1, seq = 0, gap = 0;
    static struct option lopts[] = {{"code", required_argument, 0, 1000}, {"packed", no_argument, 0, 1001},
                                    {"testframes", required_argument, 0, 1002}, {"bursts", required_argument, 0, 1003},
                                    {"source", required_argument, 0, 1004}, {"seq", no_argument, 0, 1005},
                                    {"gap", required_argument, 0, 1006}, {0, 0, 0, 0}};
    int o, oi;
    while ((o = getopt_long(argc, argv, "m:h", lopts, &oi)) != -1) {
        switch (o) {
        case 'm': M = atoi(optarg); break;
        case 1000: code_path = optarg; break;
        case 1001: packed = 1; break;
        case 1002: testframes = atoi(optarg); break;
        case 1003: bursts = atoi(optarg); break;
        case 1004: source = (int)strtol(optarg, nullptr, 0); break;
        case 1005: seq = 1; break;
        case 1006: gap = atoi(optarg); break;
        default:
            fprintf(stderr, "usage: %s --code FILE [-m 2|4] [--packed] [--testframes N [--bursts B]] [--source BYTE] [--seq] [--gap BITS] in|- out|-\n", argv[0]);
            return 1;
        }
    }
    if (argc - optind < 2 || code_path.empty() || (M != 2 && M != 4)) { fprintf(stderr, "fsk_ldpc_framer: need --code FILE, -m 2|4, input and output\n"); return 1; }
    LdpcCode code;
    const std::string err = code.load(code_path);
    if (!err.empty()) { fprintf(stderr, "fsk_ldpc_framer: %s: %s\n", code_path.c_str(), err.c_str()); return 2; }
    if (!code.accumulator) { fprintf(stderr, "fsk_ldpc_framer: %s has no dual-diagonal parity part: no linear-time encoder\n", code.name.c_str()); return 2; }
    FILE *fin = strcmp(argv[optind], "-") ? fopen(argv[optind], "rb") : stdin;
    FILE *fout = strcmp(argv[optind + 1], "-") ? fopen(argv[optind + 1], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "fsk_ldpc_framer: couldn't open files\n"); return 1; }

    const int k = code.k, bpf = code.bits_per_frame();
    fprintf(stderr, "fsk_ldpc_framer: code %s data_bits_per_frame %d bits_per_frame %d M %d\n", code.name.c_str(), k, bpf, M);
    std::vector<uint8_t> data((size_t)k), frame((size_t)bpf), zeros((size_t)(gap > 0 ? gap : 0), 0);
    const std::vector<uint8_t> pre = preamble_bits(M);

    if (testframes > 0) {
        testframe_payload(data.data(), k);
        for (int b = 0; b < bursts; b++) {
            fwrite(pre.data(), 1, pre.size(), fout);
            for (int f = 0; f < testframes; f++) {
                if (source >= 0) for (int i = 0; i < 8; i++) data[i] = (source >> (7 - i)) & 1;
                if (seq) { const int s = (f + 1) & 0xff; for (int i = 0; i < 8; i++) data[8 + i] = (s >> (7 - i)) & 1; }
                insert_crc(data.data(), k);
                frame_bits(code, data.data(), frame.data());
                fwrite(frame.data(), 1, frame.size(), fout);
            }
            if (gap > 0) fwrite(zeros.data(), 1, zeros.size(), fout);
            fprintf(stderr, "fsk_ldpc_framer: End of burst %d\n", b);
        }
    } else {
        std::vector<uint8_t> bytes((size_t)k / 8);
        int nframes = 0;
        for (;;) {
            uint8_t burst_control;
            if (fread(&burst_control, 1, 1, fin) != 1) break;
            size_t nread;
            if (packed) { nread = fread(bytes.data(), 1, bytes.size(), fin) * 8; unpack_bits_msb(data.data(), bytes.data(), k); }
            else nread = fread(data.data(), 1, (size_t)k, fin);
            if ((int)nread != k) break;
            if (burst_control == 1) { fwrite(pre.data(), 1, pre.size(), fout); nframes = 0; }
            if (burst_control == 0 || burst_control == 1) {
                insert_crc(data.data(), k);
                frame_bits(code, data.data(), frame.data());
                fwrite(frame.data(), 1, frame.size(), fout);
                nframes++;
            }
            if (burst_control == 2) {
                if (gap > 0) fwrite(zeros.data(), 1, zeros.size(), fout);
                fprintf(stderr, "fsk_ldpc_framer: Tx off after %d frames\n", nframes);
            }
            if (fout == stdout) fflush(fout);
        }
    }
    if (fout != stdout) fclose(fout);
    return 0;
}
