// fsk_mod -- bits (one per byte) -> continuous-phase M-FSK, s16 real or complex (-c).
// CLI of codec2's tool [UPSTREAM-RECALLED codec2 src/fsk_mod.c]:
//   fsk_mod [-c] [-a amp] [-t] [-p P] M Fs Rs f1 shift InputOneBitPerByte OutputModRawFile
// argv pinned by /root/reference/README.md:142,218 and script/ping:97. CPU tool (Tx side and
// synthetic-input generator of the benchmark), not on the GPU path.
#include <getopt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fsk_plan.hpp"
#include "../../include/pirip_hip.h"

int main(int argc, char **argv)
{
    int complex_out = 0, test_tone = 0, amp = PIRIP_FDMDV_SCALE, P = PIRIP_FSK_DEFAULT_P;
    int o;
    while ((o = getopt(argc, argv, "ca:tp:")) != -1) {
        switch (o) {
        case 'c': complex_out = 1; break;
        case 'a': amp = atoi(optarg); break;
        case 't': test_tone = 1; break;
        case 'p': P = atoi(optarg); break;
        default: return 1;
        }
    }
    (void)P;
    if (argc - optind < 7) {
        fprintf(stderr, "usage: %s [-c] [-a amp] [-t] [-p P] Mode SampleFreq SymbolFreq TxFreq1 TxFreqSpace InputOneBitPerByte OutputModRawFile\n", argv[0]);
        return 1;
    }
    int M = atoi(argv[optind]), Fs = atoi(argv[optind + 1]), Rs = atoi(argv[optind + 2]);
    int f1 = atoi(argv[optind + 3]), shift = atoi(argv[optind + 4]);
    FILE *fin = strcmp(argv[optind + 5], "-") ? fopen(argv[optind + 5], "rb") : stdin;
    FILE *fout = strcmp(argv[optind + 6], "-") ? fopen(argv[optind + 6], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "Couldn't open files\n"); return 1; }
    if ((M != 2 && M != 4) || Fs <= 0 || Rs <= 0 || Fs % Rs || f1 <= 0 || shift <= 0) { fprintf(stderr, "bad parameters\n"); return 1; }

    pirip::FskMod mod;
    mod.init(Fs, Rs, M, f1, shift);
    const int Nsym = PIRIP_FSK_DEFAULT_NSYM;
    const int Nbits = Nsym * (M == 2 ? 1 : 2);
    const int N = Nsym * (Fs / Rs);
    std::vector<uint8_t> bits(Nbits);
    std::vector<float> mbuf((size_t)N * 2);
    std::vector<int16_t> raw((size_t)N * 2);
    while (fread(bits.data(), 1, Nbits, fin) == (size_t)Nbits) {
        if (test_tone) memset(bits.data(), 0, Nbits);
        mod.mod(bits.data(), Nbits, mbuf.data(), complex_out);
        const int nout = complex_out ? 2 * N : N;
        // fsk_mod()'s oscillator output has peak 2.0; -a names the s16 PEAK (README.md:142 uses
        // -a 30000, which must fit int16), so scale by amp/2  [scale convention UNVERIFIED]
        for (int i = 0; i < nout; i++) raw[i] = (int16_t)(mbuf[i] * (amp / 2.0f));
        fwrite(raw.data(), sizeof(int16_t), nout, fout);
        if (fout == stdout) fflush(fout);
    }
    return 0;
}
