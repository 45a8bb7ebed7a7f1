// fsk_demod -- the reference's receive CLI, served by the HIP demodulator.
//   fsk_demod [--fsk_lower Hz] [--fsk_upper Hz] [-d|--cu8] [-c|--cs16] [-p P] [--mask spacing]
//             [-s] [--nsym N] [-t] M Fs Rs InputModemRawFile OutputOneBitPerByteFile
// [UPSTREAM-RECALLED codec2 src/fsk_demod.c]; argv forms pinned by
// /root/reference/README.md:105,109 and /root/reference/test/loopback_rtl_sdr.sh:16.
// stdin/stdout byte formats: default real s16, -c interleaved complex s16, -d interleaved
// complex u8; output one byte per bit (0/1), Nsym*log2(M) per frame, or f32 soft magnitudes
// with -s. Upstream reads exactly fsk_nin() samples per iteration; a short final read ends
// the loop and the tail is discarded -- the same frames come out here, but samples are read in
// larger chunks and handed to the GPU a chunk at a time (the unconsumed tail of a chunk is
// carried in front of the next one). On a pipe the chunk is whatever has arrived, at least one
// frame: a live source (rtl_sdr at 240 kS/s) sees its bits frame by frame like upstream's
// per-frame fflush, a fast producer (cat file |) still moves thousands of frames per GPU call.
// -t: one JSON object per frame on stderr with the modem statistics upstream's test mode prints
// for its GUI (EbNodB, ppm, the tone estimates, the eye diagram of that frame -- MODEM_STATS.rx_eye, neyetr rows of neyesamp
// points -- and "samp_fft" = the first Ndft/2 values of the estimator spectrum Sf) [UPSTREAM-RECALLED key names]. When the input is a FILE (not a pipe) the whole
// capture is there to be read: it is taken in pieces of up to 64 M samples and each piece is
// demodulated frame-parallel on many wavefronts (pirip_hip_demod_capture_host; the output is the
// read loop's, bit for bit) -- $PIRIP_FSK_DEMOD_REPORT prints how each piece went. No CPU demodulator
// exists in this program: without a HIP device it exits with an error.
#include <getopt.h>
#include <sys/ioctl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/pirip_hip.h"

int main(int argc, char **argv)
{
    {   // a binary compiled against another header generation must not run against this library (stats rows, stream state sizes)
        const int abi_ok = pirip_hip_abi_check(PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME, sizeof(pirip_stream_state));
        if (!abi_ok) { fprintf(stderr, "%s: built against a different pirip_hip.h than %s\n", argv[0], pirip_hip_version()); return 2; }
    }
    int complex_in = 0, u8_in = 0, soft = 0, P = PIRIP_FSK_DEFAULT_P, mask = 0, nsym = PIRIP_FSK_DEFAULT_NSYM;
    int user_lower = 0, user_upper = 0, fsk_lower = 0, fsk_upper = 0, testmode = 0;
    static struct option lopts[] = {
        {"fsk_lower", required_argument, 0, 'b'}, {"fsk_upper", required_argument, 0, 'u'},
        {"mask", required_argument, 0, 'm'}, {"cu8", no_argument, 0, 'd'}, {"cs16", no_argument, 0, 'c'},
        {"conv", required_argument, 0, 'p'}, {"soft-dec", no_argument, 0, 's'},
        {"nsym", required_argument, 0, 'n'}, {"testmode", no_argument, 0, 't'}, {"help", no_argument, 0, 'h'},
        {0, 0, 0, 0}};
    int o, oi;
    while ((o = getopt_long(argc, argv, "dcsp:b:u:m:n:t::h", lopts, &oi)) != -1) {
        switch (o) {
        case 'd': u8_in = 1; complex_in = 1; break;
        case 'c': complex_in = 1; break;
        case 's': soft = 1; break;
        case 'p': P = atoi(optarg); break;
        case 'b': fsk_lower = atoi(optarg); user_lower = 1; break;
        case 'u': fsk_upper = atoi(optarg); user_upper = 1; break;
        case 'm': mask = atoi(optarg); break;
        case 'n': nsym = atoi(optarg); break;
        case 't': testmode = 1; break;
        default:
            fprintf(stderr, "usage: %s [--fsk_lower Hz] [--fsk_upper Hz] [-d|-c] [-p P] [--mask spacing] [-s] M Fs Rs in out\n", argv[0]);
            return 1;
        }
    }
    if (argc - optind < 5) { fprintf(stderr, "Too few arguments\n"); return 1; }
    const int M = atoi(argv[optind]), Fs = atoi(argv[optind + 1]), Rs = atoi(argv[optind + 2]);
    FILE *fin = strcmp(argv[optind + 3], "-") ? fopen(argv[optind + 3], "rb") : stdin;
    FILE *fout = strcmp(argv[optind + 4], "-") ? fopen(argv[optind + 4], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "Couldn't open files\n"); return 1; }

    if (!user_lower) fsk_lower = complex_in ? -Fs / 2 : 0;
    if (!user_upper) fsk_upper = Fs / 2;
    fprintf(stderr, "Setting estimator limits to %d to %d Hz.\n", fsk_lower, fsk_upper);

    // real s16 input is widened to complex s16 (imag = 0) on the way in
    pirip_fsk_params prm{Fs, Rs, M, P, nsym, fsk_lower, fsk_upper, mask ? 1 : 0, mask ? mask : 100,
                         u8_in ? PIRIP_IN_CU8_FSKDEMOD : PIRIP_IN_CS16};
    // a regular file is a whole capture: frame-parallel route, the handle's streams are its work slots
    bool capture = false;
    long long file_bytes = 0;
    {
        struct stat sb;
        capture = fin != stdin && !testmode && !getenv("PIRIP_FSK_DEMOD_FRAMES") && fstat(fileno(fin), &sb) == 0 && S_ISREG(sb.st_mode);
        if (capture) file_bytes = (long long)sb.st_size;
    }
    const int slots = capture ? (getenv("PIRIP_FSK_DEMOD_SLOTS") ? atoi(getenv("PIRIP_FSK_DEMOD_SLOTS")) : 2048) : 1;
    pirip_hip_demod *h = nullptr;
    int rc = pirip_hip_create(&prm, slots > 0 ? slots : 1, -1, &h);
    if (rc != PIRIP_OK) {
        fprintf(stderr, "fsk_demod: %s (this build demodulates on an AMD GPU only; there is no CPU fallback)\n", pirip_hip_strerror(rc));
        return 2;
    }
    pirip_fsk_info info;
    pirip_hip_get_info(h, &info);
    std::vector<float> eye((size_t)8 * 160), Sf;
    if (testmode) {
        rc = pirip_hip_enable_eye(h, 1);
        if (rc != PIRIP_OK) { fprintf(stderr, "fsk_demod: %s\n", pirip_hip_strerror(rc)); return 2; }
        Sf.resize((size_t)info.Ndft);
    }

    const size_t bps_file = u8_in ? 2 : (complex_in ? 4 : 2);        // bytes per sample on the pipe
    const size_t bps_dev = (size_t)info.bytes_per_sample;             // bytes per sample on the device
    const bool is_pipe = (fin == stdin);
    if (is_pipe) setvbuf(fin, nullptr, _IONBF, 0);                     // FIONREAD below must see everything that has arrived
    // chunk: a whole number of nominal frames; small when interactive so bits flow promptly
    const char *env = getenv("PIRIP_FSK_DEMOD_FRAMES");
    long chunk_frames = env ? atol(env) : capture ? (64L << 20) / info.N : 4096;
    if (capture) {                                                     // no larger than the file
        const long file_frames = (long)(file_bytes / (long long)bps_file / info.N) + 2;
        if (file_frames < chunk_frames) chunk_frames = file_frames;
    }
    if (testmode) chunk_frames = 1;                                    // statistics are read back after every frame
    if (chunk_frames < 1) chunk_frames = 1;
    const size_t chunk = (size_t)chunk_frames * info.N;
    // frames a buffer of (carry + chunk) samples can hold when every frame is the short one (nin = N - Ts/4):
    // every call demodulates ALL whole frames present, so the carry stays below nin_max and buf never overflows
    const size_t nin_min = (size_t)(info.N - info.Ts / 4);
    const int64_t max_frames = (int64_t)((chunk + (size_t)info.nin_max) / nin_min) + 2;

    std::vector<uint8_t> buf((chunk + info.nin_max) * bps_dev);       // [carry | new]
    std::vector<uint8_t> rd(chunk * bps_file);
    std::vector<uint8_t> bits((size_t)max_frames * info.Nbits);
    std::vector<float> filt((size_t)max_frames * M * nsym);
    size_t have = 0;   // samples in buf
    for (;;) {
        // fill: at least nin samples must be present for one more frame
        size_t want = chunk;
        if (have + want > chunk + (size_t)info.nin_max) want = chunk + (size_t)info.nin_max - have;   // never past buf
        if (is_pipe && !env && !testmode) {
            // take what has arrived, but at least the samples the next frame still needs (that read blocks, like upstream's)
            int avail = 0;
            const size_t need = have < (size_t)info.nin_max ? (size_t)info.nin_max - have : 1;
            if (ioctl(fileno(fin), FIONREAD, &avail) == 0) {
                size_t a = (size_t)(avail > 0 ? avail : 0) / bps_file;
                if (a < need) a = need;
                if (a < want) want = a;
            }
        }
        size_t got = fread(rd.data(), bps_file, want, fin);
        if (complex_in || u8_in) memcpy(buf.data() + have * bps_dev, rd.data(), got * bps_file);
        else {
            int16_t *dst = (int16_t *)(buf.data() + have * bps_dev);
            const int16_t *src = (const int16_t *)rd.data();
            for (size_t i = 0; i < got; i++) { dst[2 * i] = src[i]; dst[2 * i + 1] = 0; }
        }
        have += got;
        int64_t nf = 0, cons = 0;
        if (capture) {
            pirip_capture_report rep;
            rc = pirip_hip_demod_capture_host(h, buf.data(), (int64_t)have, bits.data(), soft ? filt.data() : nullptr, nullptr, max_frames, &nf,
                                              &cons, &rep);
            if (rc == PIRIP_OK && getenv("PIRIP_FSK_DEMOD_REPORT"))
                fprintf(stderr, "capture: %lld samples -> %lld frames; %d segments of %d frames, %d pass(es), %d segment re-runs, %lld frames demodulated\n",
                        (long long)have, (long long)nf, rep.segments, rep.segment_frames, rep.passes, rep.segments_rerun, (long long)rep.frames_demodulated);
        } else
            rc = pirip_hip_demod_host(h, buf.data(), (int64_t)have, bits.data(), soft ? filt.data() : nullptr, nullptr, max_frames, &nf, &cons);
        if (rc != PIRIP_OK) { fprintf(stderr, "fsk_demod: %s\n", pirip_hip_strerror(rc)); return 2; }
        if (soft) fwrite(filt.data(), sizeof(float), (size_t)nf * M * nsym, fout);
        else fwrite(bits.data(), 1, (size_t)nf * info.Nbits, fout);
        if (fout == stdout) fflush(fout);
        if (testmode && nf > 0) {
            pirip_stream_state ss;
            if (pirip_hip_get_stream_state(h, 0, &ss) == PIRIP_OK) {
                fprintf(stderr, "{\"EbNodB\": %2.2f, \"ppm\": %d, ", ss.snr_est, (int)ss.ppm);
                for (int m = 0; m < M; m++) fprintf(stderr, "\"f%d_est\": %.1f, ", m + 1, ss.f_est[m]);
                fprintf(stderr, "\"SNRest\": %.3f, \"norm_rx_timing\": %.4f, \"eye_diagram\": [", ss.SNRest, ss.norm_rx_timing);
                int ntr = 0, npt = 0;
                if (pirip_hip_get_eye(h, 0, 1, eye.data(), &ntr, &npt) != PIRIP_OK) ntr = 0;
                for (int i = 0; i < ntr; i++) {
                    fprintf(stderr, "[");
                    for (int j = 0; j < npt; j++) fprintf(stderr, "%f%s", eye[(size_t)i * 160 + j], j + 1 < npt ? ", " : "");
                    fprintf(stderr, "]%s", i + 1 < ntr ? ", " : "");
                }
                fprintf(stderr, "], \"samp_fft\": [");
                if (pirip_hip_get_Sf(h, 0, Sf.data()) == PIRIP_OK)
                    for (int i = 0; i < info.Ndft / 2; i++) fprintf(stderr, "%f%s", Sf[(size_t)i], i + 1 < info.Ndft / 2 ? ", " : "");
                fprintf(stderr, "]}\n");
            }
        }
        memmove(buf.data(), buf.data() + (size_t)cons * bps_dev, (have - (size_t)cons) * bps_dev);
        have -= (size_t)cons;
        if (got < want) break;   // EOF: the tail shorter than nin is discarded, as upstream
    }
    pirip_hip_destroy(h);
    if (fout != stdout) fclose(fout);
    return 0;
}
