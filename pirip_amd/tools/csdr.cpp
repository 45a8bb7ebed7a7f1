// csdr -- the three csdr sub-commands pirip's wide-band receive pipe uses, on the GPU:
//   csdr convert_u8_f | csdr fir_decimate_cc D [transition_bw [HAMMING]] | csdr convert_f_s16
// (/root/reference/README.md:109,162; behaviour UPSTREAM-RECALLED from ha7ilm/csdr csdr.c).
// stdin/stdout carry raw samples exactly as csdr's: u8 -> f32 -> f32 -> s16.
// fir_decimate_cc keeps csdr's block structure so the number of samples it emits for a finite
// input matches: blocks of `bufsize` complex samples (16384), the taps are zero-padded to a
// multiple of 4 for the "enough input left" test, consumed = D*outputs, a short read ends the
// stream and the trailing partial block is dropped.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/pirip_hip.h"

static int badsyntax(const char *why) { fprintf(stderr, "csdr: %s\n", why); return -1; }

int main(int argc, char **argv)
{
    {   // a binary compiled against another header generation must not run against this library (stats rows, stream state sizes)
        const int abi_ok = pirip_hip_abi_check(PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME, sizeof(pirip_stream_state));
        if (!abi_ok) { fprintf(stderr, "%s: built against a different pirip_hip.h than %s\n", argv[0], pirip_hip_version()); return 2; }
    }
    if (argc < 2) return badsyntax("need a function name (convert_u8_f | fir_decimate_cc | convert_f_s16)");
    if (pirip_hip_device_count() <= 0) {
        fprintf(stderr, "csdr: no usable HIP device (this build runs on an AMD GPU only; there is no CPU fallback)\n");
        return 2;
    }
    const int bufsize = 16384;
    if (!strcmp(argv[1], "convert_u8_f")) {
        const int n = bufsize * 64;
        std::vector<unsigned char> in(n); std::vector<float> out(n);
        size_t got;
        while ((got = fread(in.data(), 1, n, stdin)) > 0) {
            convert_u8_f(in.data(), out.data(), (int)got);
            fwrite(out.data(), sizeof(float), got, stdout); fflush(stdout);
        }
        return 0;
    }
    if (!strcmp(argv[1], "convert_f_s16")) {
        const int n = bufsize * 64;
        std::vector<float> in(n); std::vector<short> out(n);
        size_t got;
        while ((got = fread(in.data(), sizeof(float), n, stdin)) > 0) {
            convert_f_s16(in.data(), out.data(), (int)got);
            fwrite(out.data(), sizeof(short), got, stdout); fflush(stdout);
        }
        return 0;
    }
    if (!strcmp(argv[1], "fir_decimate_cc")) {
        if (argc <= 2) return badsyntax("need required parameter (decimation factor)");
        int factor = atoi(argv[2]);
        float tbw = 0.05f;
        if (argc >= 4) tbw = atof(argv[3]);
        if (argc >= 5 && strcasecmp(argv[4], "HAMMING")) return badsyntax("only the HAMMING window (csdr's default) is built");
        if (factor < 1) return badsyntax("bad decimation factor");
        int L = firdes_filter_len(tbw);
        fprintf(stderr, "fir_decimate_cc: window = HAMMING\nfir_decimate_cc: taps_length = %d\n", L);
        int the_bufsize = bufsize;
        while (the_bufsize < L * 2) the_bufsize *= 2;
        const int Lp = L + 3 - ((L + 3) % 4);
        std::vector<float> taps(Lp, 0.f);
        firdes_lowpass_f_hamming(taps.data(), L, 0.5 / (float)factor);
        // blocks per GPU call: csdr's block loop is replayed K blocks at a time
        const char *env = getenv("PIRIP_CSDR_BLOCKS");
        int K = env ? atoi(env) : 64; if (K < 1) K = 1;
        const int osz = (the_bufsize - Lp) / factor + 1;     // outputs per full block
        const int skip = osz * factor;                        // samples consumed per block
        std::vector<complexf> ibuf((size_t)the_bufsize + (size_t)(K - 1) * skip);
        std::vector<complexf> obuf((size_t)K * osz);
        if (fread(ibuf.data(), sizeof(complexf), the_bufsize, stdin) != (size_t)the_bufsize) return 0;
        for (;;) {
            // ibuf holds one full block; try to extend it by up to K-1 further block advances
            size_t extra = fread(ibuf.data() + the_bufsize, sizeof(complexf), (size_t)(K - 1) * skip, stdin);
            int nblk = 1 + (int)(extra / skip);
            bool eof = extra < (size_t)(K - 1) * skip;
            size_t span = (size_t)the_bufsize + (size_t)(nblk - 1) * skip;
            int n = fir_decimate_cc(ibuf.data(), obuf.data(), (int)span, factor, taps.data(), Lp);
            if (n != nblk * osz) { fprintf(stderr, "csdr: GPU decimator returned %d outputs, expected %d\n", n, nblk * osz); return 2; }
            fwrite(obuf.data(), sizeof(complexf), n, stdout); fflush(stdout);
            if (eof) return 0;   // the partial block after the last full advance is dropped
            // next block starts nblk*skip in; carry the overlap and read one block advance
            memmove(ibuf.data(), ibuf.data() + (size_t)nblk * skip, (span - (size_t)nblk * skip) * sizeof(complexf));
            size_t need = (size_t)the_bufsize - (span - (size_t)nblk * skip);
            if (fread(ibuf.data() + (span - (size_t)nblk * skip), sizeof(complexf), need, stdin) != need) return 0;
        }
    }
    return badsyntax("function name given in argument 1 does not exist in this build");
}
