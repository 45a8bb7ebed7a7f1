// fsk_put_test_bits -- count bit errors against the 100-bit test frame; exit code = verdict.
// CLI of codec2's tool [UPSTREAM-RECALLED codec2 src/fsk_put_test_bits.c]:
//   fsk_put_test_bits [-b berPass] [-p packetsPass] [-t validFrameBER] [-f frameBits] [-q] InputOneBitPerByte
// `-q` and `-p` are the forms the reference exercises: /root/reference/test/loopback_rtl_fsk.sh:10,
// test/loopback_rtl_sdr.sh:16; bare form /root/reference/README.md:105. CPU tool.
#include <getopt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "fsk_plan.hpp"

int main(int argc, char **argv)
{
    int framesize = 100, packet_pass = 0, quiet = 0;
    float valid_thresh = 0.1f, ber_pass = 0.0f;
    int o;
    while ((o = getopt(argc, argv, "b:p:t:f:qh")) != -1) {
        switch (o) {
        case 'b': ber_pass = atof(optarg); break;
        case 'p': packet_pass = atoi(optarg); break;
        case 't': valid_thresh = atof(optarg); break;
        case 'f': framesize = atoi(optarg); break;
        case 'q': quiet = 1; break;
        default:
            fprintf(stderr, "usage: %s [-b berPass] [-p packetsPass] [-t validBER] [-f frameBits] [-q] InputOneBitPerByte\n", argv[0]);
            return 1;
        }
    }
    if (optind >= argc) { fprintf(stderr, "Too few arguments\n"); return 1; }
    FILE *fin = strcmp(argv[optind], "-") ? fopen(argv[optind], "rb") : stdin;
    if (!fin) { fprintf(stderr, "Couldn't open input file: %s\n", argv[optind]); return 1; }
    // ber_pass stays 0 unless -b is given [UPSTREAM-RECALLED: ber_pass_thresh defaults to 0], so `-q -p N`
    // (/root/reference/test/loopback_rtl_sdr.sh:16) passes only with N valid packets AND no bit errors in them

    pirip::PutBits pb;
    pb.init(framesize, valid_thresh);
    int c;
    while ((c = fgetc(fin)) != EOF) {
        int errs;
        if (pb.push((uint8_t)c, &errs) && !quiet)
            fprintf(stderr, "[%04d] BER %5.3f, bits tested %6ld, bit errors %6ld errs: %4d \n",
                    pb.packetcnt, pb.ber(), pb.bitcnt, pb.biterr, errs);
    }
    const float ber = pb.ber();
    fprintf(stderr, "[%04d] BER %5.3f, bits tested %6ld, bit errors %6ld\n", pb.packetcnt, ber, pb.bitcnt, pb.biterr);
    if (pb.packetcnt >= packet_pass && pb.bitcnt > 0 && ber <= ber_pass) { fprintf(stderr, "PASS\n"); return 0; }
    fprintf(stderr, "FAIL\n");
    return 1;
}
