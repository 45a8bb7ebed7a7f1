// rtl_fsk -- pirip's integrated receiver (in-process convert_u8_f [-> fir_decimate_cc] -> fsk_demod),
// served by the HIP path. [UPSTREAM-RECALLED drowe67/librtlsdr src/rtl_fsk.c, branch development,
// cloned un-pinned by /root/reference/build_rtlsdr.sh:4-10.]
//
// argv surface kept from the reference's command lines:
//   /root/reference/test/loopback_rtl_fsk.sh:10   rtl_fsk -g 1 -s $Fs -f $rx_freq - -n N -u host
//   /root/reference/README.md:114,123,152,172     -w bw -e gains -r Rs -a modemFs -m M --mask S
//   /root/reference/script/frame_repeater:23,36,43 and script/ping:47   --code NAME --filter A -q -b -L -v
// There is no dongle on a GPU node, so the 8-bit IQ comes from `-i FILE|-` (documented
// extension); tuner options (-g -f -w -e -p) are accepted and ignored.
// Not built: `--code` (FSK_LDPC needs codec2's H_256_512_4 tables, absent: SURVEY.md 8f-1).
//
// Behaviour: u8 IQ at the RTL rate -s (default 240000); if -a modemFs differs, decimate by
// rtlFs/modemFs with csdr's windowed-sinc (section B of pirip_hip.h) and demodulate cs16 at the
// modem rate; else demodulate the u8 directly with csdr's x/127.5-1 conversion. One byte per bit
// on the output ("-" = stdout). -u host: once per second of samples one JSON line to UDP
// host:8001 with the keys script/dash.py reads (/root/reference/script/dash.py:26-45).
#include <arpa/inet.h>
#include <getopt.h>
#include <netdb.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/pirip_hip.h"

static void usage()
{
    fprintf(stderr,
            "rtl_fsk (pirip_hip): -i <u8 IQ file|-> [-s rtlFs] [-a modemFs] [-r Rs] [-m M] [-n nSamples]\n"
            "        [--mask spacing] [-l fsk_lower] [-U fsk_upper] [-u dashHost] [-v] [-q] <out|->\n"
            "        (tuner options -g -f -w -e -p are accepted and ignored; --code is not built)\n");
}

int main(int argc, char **argv)
{
    long rtlFs = 240000, modemFs = 0, Rs = 10000, nsamples = 0;
    int M = 2, mask = 0, verbose = 0, fsk_lower = 0, fsk_upper = 0, user_lower = 0, user_upper = 0;
    std::string in_name, dash_host, code;
    static struct option lopts[] = {{"code", required_argument, 0, 1000}, {"mask", required_argument, 0, 1001},
                                    {"filter", required_argument, 0, 1002}, {"testframes", no_argument, 0, 1003},
                                    {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    int o, oi;
    while ((o = getopt_long(argc, argv, "i:g:s:f:n:u:r:a:w:e:m:p:l:U:vqbLh", lopts, &oi)) != -1) {
        switch (o) {
        case 'i': in_name = optarg; break;
        case 's': rtlFs = (long)atof(optarg); break;
        case 'a': modemFs = (long)atof(optarg); break;
        case 'r': Rs = (long)atof(optarg); break;
        case 'm': M = atoi(optarg); break;
        case 'n': nsamples = (long)atof(optarg); break;
        case 'u': dash_host = optarg; break;
        case 'l': fsk_lower = atoi(optarg); user_lower = 1; break;
        case 'U': fsk_upper = atoi(optarg); user_upper = 1; break;
        case 'v': verbose = 1; break;
        case 'g': case 'f': case 'w': case 'e': case 'p': case 'q': case 'b': case 'L': break;   // tuner / log options
        case 1000: code = optarg; break;
        case 1001: mask = atoi(optarg); break;
        case 1002: case 1003: break;
        default: usage(); return 1;
        }
    }
    if (optind >= argc) { usage(); return 1; }
    if (!code.empty()) {
        fprintf(stderr, "rtl_fsk: --code %s needs codec2's LDPC tables, which this build does not have (SURVEY.md 8f-1)\n", code.c_str());
        return 2;
    }
    if (in_name.empty()) {
        fprintf(stderr, "rtl_fsk: no RTL-SDR hardware support in this build; give the 8-bit IQ with -i FILE or -i -\n");
        return 2;
    }
    FILE *fin = in_name == "-" ? stdin : fopen(in_name.c_str(), "rb");
    FILE *fout = strcmp(argv[optind], "-") ? fopen(argv[optind], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "rtl_fsk: couldn't open files\n"); return 1; }
    if (!modemFs) modemFs = rtlFs;
    if (rtlFs % modemFs) { fprintf(stderr, "rtl_fsk: rtl rate must be a multiple of the modem rate\n"); return 1; }
    const int D = (int)(rtlFs / modemFs);
    const int Fs = (int)modemFs;
    if (Fs % Rs) { fprintf(stderr, "rtl_fsk: modem rate must be a multiple of the symbol rate\n"); return 1; }
    int Ts = Fs / (int)Rs, P = Ts;
    while (P > 10 && (P % 2) == 0) P /= 2;        // oversample reduction rule [UPSTREAM-RECALLED, unverified]
    if (P < 4) P = Ts;
    if (!user_lower) fsk_lower = (int)Rs / 2;     // keep the estimator off the dongle's DC spur (README.md:116)
    if (!user_upper) fsk_upper = Fs / 2;

    pirip_fsk_params prm{Fs, (int)Rs, M, P, PIRIP_FSK_DEFAULT_NSYM, fsk_lower, fsk_upper, mask ? 1 : 0,
                         mask ? mask : 100, D > 1 ? PIRIP_IN_CS16 : PIRIP_IN_CU8_CSDR};
    pirip_hip_demod *h = nullptr;
    int rc = pirip_hip_create(&prm, 1, -1, &h);
    if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: %s (AMD GPU only; there is no CPU fallback)\n", pirip_hip_strerror(rc)); return 2; }
    pirip_hip_decim *dec = nullptr;
    if (D > 1 && (rc = pirip_hip_decim_create(D, 0.05f, 1, -1, &dec)) != PIRIP_OK) {
        fprintf(stderr, "rtl_fsk: decimator: %s\n", pirip_hip_strerror(rc)); return 2;
    }
    pirip_fsk_info info;
    pirip_hip_get_info(h, &info);
    fprintf(stderr, "rtl_fsk: Fs %d Rs %ld M %d P %d decimation %d estimator %d..%d Hz\n", Fs, Rs, M, P, D, fsk_lower, fsk_upper);

    int sock = -1; sockaddr_in dst{};
    if (!dash_host.empty()) {
        hostent *he = gethostbyname(dash_host.c_str());
        if (he) {
            sock = socket(AF_INET, SOCK_DGRAM, 0);
            dst.sin_family = AF_INET; dst.sin_port = htons(8001);
            memcpy(&dst.sin_addr, he->h_addr_list[0], he->h_length);
        } else fprintf(stderr, "rtl_fsk: can't resolve %s, dashboard output off\n", dash_host.c_str());
    }

    // block loop: ~0.25 s of RTL samples per GPU call
    const size_t blk = (size_t)rtlFs / 4;
    std::vector<uint8_t> raw(2 * (blk + 4096 + (size_t)D * 128)), lo;     // [carry | new] at the RTL rate
    std::vector<int16_t> modem;                                            // [carry | new] at the modem rate (D > 1)
    size_t raw_have = 0, modem_have = 0;
    const int64_t max_frames = (int64_t)(blk / D / info.N) + 8;
    std::vector<uint8_t> bits((size_t)max_frames * info.Nbits);
    std::vector<float> stats((size_t)max_frames * PIRIP_STATS_PER_FRAME), Sf(info.Ndft);
    std::vector<float> timing_acc;
    void *d_raw = nullptr, *d_dec = nullptr;
    if (D > 1) {
        if (hipMalloc(&d_raw, raw.size()) != hipSuccess || hipMalloc(&d_dec, 4 * (raw.size() / 2 / D + 16)) != hipSuccess) {
            fprintf(stderr, "rtl_fsk: hipMalloc failed\n"); return 2;
        }
        modem.resize(2 * (blk / D + 4096 + info.nin_max));
    }
    long total_in = 0, since_json = 0, frame_no = 0;
    for (;;) {
        size_t want = blk;
        if (nsamples && total_in + (long)want > nsamples) want = (size_t)(nsamples - total_in);
        size_t got = want ? fread(raw.data() + 2 * raw_have, 2, want, fin) : 0;
        total_in += (long)got; raw_have += got;
        int64_t nf = 0, cons = 0;
        if (D == 1) {
            rc = pirip_hip_demod_host(h, raw.data(), (int64_t)raw_have, bits.data(), nullptr, stats.data(), max_frames, &nf, &cons);
            if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: %s\n", pirip_hip_strerror(rc)); return 2; }
            memmove(raw.data(), raw.data() + 2 * cons, 2 * (raw_have - (size_t)cons));
            raw_have -= (size_t)cons;
        } else {
            const int64_t nout = pirip_hip_decim_nout(dec, (int64_t)raw_have);
            if (nout > 0) {
                if (hipMemcpy(d_raw, raw.data(), 2 * raw_have, hipMemcpyHostToDevice) != hipSuccess) return 2;
                rc = pirip_hip_decim_batch(dec, (const uint8_t *)d_raw, 0, (int64_t)raw_have, d_dec, 0, 1, nullptr);
                if (rc != PIRIP_OK || hipMemcpy(modem.data() + 2 * modem_have, d_dec, 4 * (size_t)nout, hipMemcpyDeviceToHost) != hipSuccess) {
                    fprintf(stderr, "rtl_fsk: decimator failed\n"); return 2;
                }
                modem_have += (size_t)nout;
                const size_t used = (size_t)nout * D;           // overlap carry: consumed = D * outputs
                memmove(raw.data(), raw.data() + 2 * used, 2 * (raw_have - used));
                raw_have -= used;
            }
            rc = pirip_hip_demod_host(h, modem.data(), (int64_t)modem_have, bits.data(), nullptr, stats.data(), max_frames, &nf, &cons);
            if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: %s\n", pirip_hip_strerror(rc)); return 2; }
            memmove(modem.data(), modem.data() + 2 * cons, 4 * (modem_have - (size_t)cons));
            modem_have -= (size_t)cons;
        }
        fwrite(bits.data(), 1, (size_t)nf * info.Nbits, fout);
        if (fout == stdout) fflush(fout);
        for (int64_t f = 0; f < nf; f++) {
            const float *s = &stats[(size_t)f * PIRIP_STATS_PER_FRAME];
            timing_acc.push_back(s[4]);
            if (verbose) fprintf(stderr, "%ld nbits: %d snr_lin: %.2f timing: %+.3f f_est: %.0f %.0f\n", frame_no, info.Nbits, s[5], s[4], s[0], s[1]);
            frame_no++;
        }
        since_json += (long)got;
        if (sock >= 0 && nf > 0 && since_json >= rtlFs) {
            since_json = 0;
            pirip_hip_get_Sf(h, 0, Sf.data());
            const float *s = &stats[(size_t)(nf - 1) * PIRIP_STATS_PER_FRAME];
            std::string js = "{\"SNRest_lin\": " + std::to_string(s[5]) + ", \"norm_rx_timing\": [";
            for (size_t i = 0; i < timing_acc.size(); i++) js += (i ? ", " : "") + std::to_string(timing_acc[i]);
            js += "], \"SfdB\": [";
            for (int i = 0; i < info.Ndft; i++) js += (i ? ", " : "") + std::to_string(20.0 * log10(Sf[i] + 1e-12));
            js += "], \"fsk_lower_Hz\": " + std::to_string(fsk_lower) + ", \"fsk_upper_Hz\": " + std::to_string(fsk_upper) + ", \"f_est_Hz\": [";
            for (int m = 0; m < M; m++) js += (m ? ", " : "") + std::to_string(s[m]);
            js += "], \"Fs_Hz\": " + std::to_string(Fs) + "}\n";
            sendto(sock, js.data(), js.size(), 0, (sockaddr *)&dst, sizeof(dst));
            timing_acc.clear();
        }
        if (got < blk) break;
    }
    pirip_hip_destroy(h);
    if (dec) pirip_hip_decim_destroy(dec);
    if (d_raw) (void)hipFree(d_raw);
    if (d_dec) (void)hipFree(d_dec);
    return 0;
}
