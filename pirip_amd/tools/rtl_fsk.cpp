// rtl_fsk -- pirip's integrated receiver (in-process convert_u8_f [-> fir_decimate_cc] -> fsk_demod [-> FSK_LDPC rx]),
// served by the HIP path. [UPSTREAM-RECALLED drowe67/librtlsdr src/rtl_fsk.c, branch development, cloned un-pinned by
// /root/reference/build_rtlsdr.sh:4-10.]
//
// argv surface kept from the reference's command lines:
//   /root/reference/test/loopback_rtl_fsk.sh:10   rtl_fsk -g 1 -s $Fs -f $rx_freq - -n N -u host
//   /root/reference/README.md:114,123,152,172     -w bw -e gains -r Rs -a modemFs -m M --mask S
//   /root/reference/README.md:184,196,239,262,286,292,297   --code NAME -v --testframes --filter A -q
//   /root/reference/script/frame_repeater:23,36,43 and script/ping:47   --code NAME --filter A -q -b -L -v
// There is no dongle on a GPU node: the 8-bit IQ comes from `-i FILE|-` or, so that the reference's command lines run
// unchanged, from the file named by the environment variable PIRIP_IQ_FILE (documented extension);
// tuner options (-g -f -w -e -p) are accepted and ignored.
// RTL sample rate (the rate the IQ file must be at): `-s` when given; else 240000 when the modem rate `-a` is absent or
// divides it (README.md:114,152,184,239; script/ping:47 -a 40000); else 1800000 -- the rate README.md:109 runs the dongle
// at, and the common multiple of the README's other modem rates (-a 200000 / 180000 / 100000: README.md:196,262,286,292,297);
// else the smallest multiple of the modem rate the dongle can do (> 900 kS/s). An RTL2832 cannot sample at 100-200 kS/s, so
// upstream must pick a multiple too; which one is [UPSTREAM-RECALLED, unverified] -- it only fixes the rate of the test file.
//
// Data path, all on the device between the upload of a block of u8 IQ and the download of bits / records:
//   u8 IQ --(rtlFs != modemFs: csdr's windowed-sinc decimator, complex float out, pirip_hip.h section B)--> modem-rate
//   samples --> pirip_hip_demod_batch (bits, per-frame stats), or with --code pirip_hip_fsk_ldpc_rx_batch (section E: the
//   demodulator hands bit LLRs to the FSK_LDPC receiver on the device) --> status / payload records.
// Output ("-" = stdout):
//   uncoded            one byte per bit, Nsym*log2(M) per demodulator call
//   --code NAME        packed payload bytes (k/8 per frame) of every frame whose CRC16 matches (README.md:297 `| hexdump`)
//   --code NAME -b     for EVERY demodulator call one rx_status byte + k/8 data bytes, zeros when no frame
//                      (/root/reference/tx/frame_repeater.c:55-62; status bits :71,80,88)
//   --filter A         frames whose first byte (source address) equals A are dropped (README.md:303-304)
// -v: one line per decoded frame in the reference's format (README.md:200-208): the first column counts frame periods
// (bits_per_frame bits of demodulator output each) since start, `nbits` is what is left in the current period and cycles the
// way the README's log shows (+6 per frame at 50 bits per call, +56 at 100), `uw_loc` stays put while in sync.
// -L: one line per frame with a good CRC on stderr -- "Terminal 1 logs metadata for each frame (Signal and Noise Power,
// SNR, time of arrival)" (README.md:59; script/ping:47 redirects stderr into /var/log/ping) -- source / sequence byte, S, N
// (struct FSK rx_sig_pow / rx_nse_pow of that demodulator call), SNR in dB, arrival time on the wall clock and on the
// sample clock [UPSTREAM-RECALLED: the exact wording of upstream's line is unverified; the keys are README.md:59's].
// -u host: once per second of samples one JSON line to UDP host:8001 with the keys script/dash.py reads
// (/root/reference/script/dash.py:26-45).
// --code NAME resolves to NAME as a file path, then $PIRIP_CODE_DIR/NAME.code, then <exe>/../data/NAME.code: codec2's
// H_256_512_4 table is not in /root/reference, it is a data drop in the format of csrc/fsk_ldpc.hpp.
#include <arpa/inet.h>
#include <getopt.h>
#include <netdb.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/pirip_hip.h"
#include "fsk_ldpc.hpp"

// The two rules of upstream's rtl_fsk.c this tool holds from recall, as data (the tool-level part of the pin-day drill; the
// demodulator's own recalled constants are pirip_fsk_recalled / PIRIP_RECALLED): flipped without a rebuild through
//   PIRIP_RTL_FSK_RULES="p_rule=0|1|2,p_max=10,default_rate=240000,wide_rate=1800000,min_rate=900001"
struct RtlFskRules {
    int p_rule = 0;                 // timing oversample: 0: halve Ts while it is > p_max and even (recalled); 1: P = Ts; 2: P = 8 (fsk_demod's default)
    int p_max = 10;
    long default_rate = 240000;     // RTL rate when -s is absent and the modem rate is absent or divides it
    long wide_rate = 1800000;       // ... else this one when the modem rate divides it
    long min_rate = 900001;         // ... else the smallest multiple of the modem rate from here up
    bool from_env()
    {
        const char *e = getenv("PIRIP_RTL_FSK_RULES");
        if (!e) return true;
        std::string all(e);
        for (size_t pos = 0; pos < all.size();) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string tok = all.substr(pos, end - pos);
            pos = end + 1;
            const size_t eq = tok.find('=');
            if (eq == std::string::npos) return false;
            const std::string k = tok.substr(0, eq);
            const long v = atol(tok.c_str() + eq + 1);
            if (k == "p_rule") p_rule = (int)v; else if (k == "p_max") p_max = (int)v; else if (k == "default_rate") default_rate = v;
            else if (k == "wide_rate") wide_rate = v; else if (k == "min_rate") min_rate = v; else return false;
        }
        return p_rule >= 0 && p_rule <= 2 && p_max >= 4 && default_rate > 0 && wide_rate > 0 && min_rate > 0;
    }
};

static void usage()
{
    fprintf(stderr,
            "rtl_fsk (pirip_hip): [-i <u8 IQ file|->] [-s rtlFs] [-a modemFs] [-r Rs] [-m M] [-n nSamples] [--mask spacing]\n"
            "        [-l fsk_lower] [-U fsk_upper] [--code NAME|FILE [--filter addr] [--testframes] [-b]] [-u dashHost] [-v] [-q] <out|->\n"
            "        IQ source: -i, or the file named by $PIRIP_IQ_FILE; tuner options -g -f -w -e -p are accepted and ignored\n");
}

static bool file_exists(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

static std::string resolve_code(const std::string &name, const char *argv0)
{
    if (file_exists(name)) return name;
    if (const char *d = getenv("PIRIP_CODE_DIR")) { const std::string p = std::string(d) + "/" + name + ".code"; if (file_exists(p)) return p; }
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    std::string base = n > 0 ? std::string(exe, (size_t)n) : std::string(argv0);
    const size_t s = base.rfind('/');
    base = s == std::string::npos ? "." : base.substr(0, s);
    const std::string p = base + "/../data/" + name + ".code";
    return file_exists(p) ? p : std::string();
}

#define HIPOK(expr) do { if ((expr) != hipSuccess) { fprintf(stderr, "rtl_fsk: HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
    {   // a binary compiled against another header generation must not run against this library (stats rows, stream state sizes)
        const int abi_ok = pirip_hip_abi_check(PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME, sizeof(pirip_stream_state));
        if (!abi_ok) { fprintf(stderr, "%s: built against a different pirip_hip.h than %s\n", argv[0], pirip_hip_version()); return 2; }
    }
    long rtlFs = 0, modemFs = 0, Rs = 10000, nsamples = 0;
    int M = 2, mask = 0, verbose = 0, quiet = 0, fsk_lower = 0, fsk_upper = 0, user_lower = 0, user_upper = 0, log_frames = 0;
    int status_bytes = 0, testframes = 0, filter = -1;
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
        case 'q': quiet = 1; break;
        case 'b': status_bytes = 1; break;
        case 'L': log_frames = 1; break;
        case 'g': case 'f': case 'w': case 'e': case 'p': break;             // tuner options
        case 1000: code = optarg; break;
        case 1001: mask = atoi(optarg); break;
        case 1002: filter = (int)strtol(optarg, nullptr, 0); break;
        case 1003: testframes = 1; break;
        default: usage(); return 1;
        }
    }
    if (optind >= argc) { usage(); return 1; }
    if (in_name.empty()) { if (const char *e = getenv("PIRIP_IQ_FILE")) in_name = e; }
    if (in_name.empty()) {
        fprintf(stderr, "rtl_fsk: no RTL-SDR hardware support in this build; give the 8-bit IQ with -i FILE, -i - or $PIRIP_IQ_FILE\n");
        return 2;
    }
    std::string code_path;
    if (!code.empty()) {
        code_path = resolve_code(code, argv[0]);
        if (code_path.empty()) {
            fprintf(stderr, "rtl_fsk: no table for --code %s: codec2's LDPC tables are not part of this build (SURVEY.md 7.6);\n"
                            "         drop %s.code (format: pirip_amd/csrc/fsk_ldpc.hpp) into $PIRIP_CODE_DIR or pass a file path\n",
                    code.c_str(), code.c_str());
            return 2;
        }
    } else if (status_bytes || testframes || filter >= 0) {
        fprintf(stderr, "rtl_fsk: -b / --testframes / --filter need --code\n");
        return 1;
    }
    FILE *fin = in_name == "-" ? stdin : fopen(in_name.c_str(), "rb");
    FILE *fout = strcmp(argv[optind], "-") ? fopen(argv[optind], "wb") : stdout;
    if (!fin || !fout) { fprintf(stderr, "rtl_fsk: couldn't open files\n"); return 1; }
    if (modemFs < 0 || rtlFs < 0 || Rs <= 0) { usage(); return 1; }
    RtlFskRules rules;
    if (!rules.from_env()) { fprintf(stderr, "rtl_fsk: PIRIP_RTL_FSK_RULES: unknown key or value out of range\n"); return 2; }
    if (!rtlFs) {                                     // no -s: see the header for the rule
        if (!modemFs || rules.default_rate % modemFs == 0) rtlFs = rules.default_rate;
        else if (rules.wide_rate % modemFs == 0) rtlFs = rules.wide_rate;
        else rtlFs = modemFs * ((rules.min_rate + modemFs - 1) / modemFs);
    }
    if (!modemFs) modemFs = rtlFs;
    if (rtlFs % modemFs) { fprintf(stderr, "rtl_fsk: rtl rate %ld must be a multiple of the modem rate %ld\n", rtlFs, modemFs); return 1; }
    const int D = (int)(rtlFs / modemFs);
    const int Fs = (int)modemFs;
    if (Fs % Rs) { fprintf(stderr, "rtl_fsk: modem rate must be a multiple of the symbol rate\n"); return 1; }
    int Ts = Fs / (int)Rs, P = Ts;
    if (rules.p_rule == 0) while (P > rules.p_max && (P % 2) == 0) P /= 2;        // oversample reduction rule [UPSTREAM-RECALLED, unverified: RtlFskRules]
    else if (rules.p_rule == 2 && Ts % 8 == 0) P = 8;
    if (P < 4) P = Ts;
    if (!user_lower) fsk_lower = (int)Rs / 2;     // keep the estimator off the dongle's DC spur (README.md:116)
    if (!user_upper) fsk_upper = Fs / 2;

    // the in-process decimator hands complex float to the modem (no s16 hop inside rtl_fsk)
    pirip_fsk_params prm{Fs, (int)Rs, M, P, PIRIP_FSK_DEFAULT_NSYM, fsk_lower, fsk_upper, mask ? 1 : 0,
                         mask ? mask : 100, D > 1 ? PIRIP_IN_CF32 : PIRIP_IN_CU8_CSDR};
    pirip_hip_demod *h = nullptr;
    int rc = pirip_hip_create(&prm, 1, -1, &h);
    if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: %s (AMD GPU only; there is no CPU fallback)\n", pirip_hip_strerror(rc)); return 2; }
    pirip_hip_decim *dec = nullptr;
    if (D > 1 && (rc = pirip_hip_decim_create(D, 0.05f, 0, -1, &dec)) != PIRIP_OK) {
        fprintf(stderr, "rtl_fsk: decimator: %s\n", pirip_hip_strerror(rc)); return 2;
    }
    pirip_fsk_info info;
    pirip_hip_get_info(h, &info);
    pirip_hip_ldpc *ldpc = nullptr;
    pirip_ldpc_info li{};
    pirip::LdpcCode tfcode;                          // --testframes: the known payload, for the ecdd column
    std::vector<uint8_t> tf_bytes;
    if (!code_path.empty()) {
        rc = pirip_hip_ldpc_create(code_path.c_str(), M, PIRIP_FSK_DEFAULT_NSYM, 1, -1, &ldpc);
        if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: --code %s: %s\n", code_path.c_str(), pirip_hip_strerror(rc)); return 2; }
        pirip_hip_ldpc_get_info(ldpc, &li);
        if (testframes && tfcode.load(code_path).empty()) {
            std::vector<uint8_t> bits((size_t)li.k);
            pirip::testframe_payload(bits.data(), li.k);
            tf_bytes.resize((size_t)li.data_bytes);
            pirip::pack_bits_msb(tf_bytes.data(), bits.data(), li.k);
        }
    }
    if (!quiet || getenv("PIRIP_RTL_FSK_BANNER"))          // (the environment switch lets a test read the banner of a -q command line)
        fprintf(stderr, "rtl_fsk: rtl rate %ld Fs %d Rs %ld M %d P %d decimation %d estimator %d..%d Hz kernel %s%s%s\n", rtlFs, Fs, Rs, M, P, D,
                fsk_lower, fsk_upper, pirip_hip_get_kernel(h) == PIRIP_KERNEL_WAVE ? "wave" : pirip_hip_get_kernel(h) == PIRIP_KERNEL_BLOCK ? "block" : "general", ldpc ? " code " : "", ldpc ? li.name : "");

    int sock = -1; sockaddr_in dst{};
    if (!dash_host.empty()) {
        hostent *he = gethostbyname(dash_host.c_str());
        if (he) {
            sock = socket(AF_INET, SOCK_DGRAM, 0);
            dst.sin_family = AF_INET; dst.sin_port = htons(8001);
            memcpy(&dst.sin_addr, he->h_addr_list[0], he->h_length);
        } else fprintf(stderr, "rtl_fsk: can't resolve %s, dashboard output off\n", dash_host.c_str());
    }

    // block loop: ~0.25 s of RTL samples per GPU round trip. Buffer sizes come from the worst-case carry:
    //   raw (D > 1): what the decimator leaves is < taps + D samples;  modem: what the demodulator leaves is < nin_max samples
    const int taps_pad = 128;                                   // >= csdr's padded tap count (79 -> 80)
    const size_t blk = (size_t)rtlFs / 4 > (size_t)(2 * info.nin_max * D) ? (size_t)rtlFs / 4 : (size_t)(2 * info.nin_max * D);
    const size_t raw_cap = blk + (size_t)(D > 1 ? taps_pad + D : info.nin_max) + 16;
    const size_t mod_cap = D > 1 ? blk / D + (size_t)info.nin_max + 16 : raw_cap;
    const size_t bps_mod = (size_t)info.bytes_per_sample;       // 8 (complex float) behind the decimator, 2 (u8) direct
    const size_t nin_min = (size_t)(info.N - info.Ts / 4);
    const int64_t max_frames = (int64_t)(mod_cap / nin_min) + 2;
    std::vector<uint8_t> raw(2 * raw_cap);
    size_t raw_have = 0, mod_have = 0;
    void *d_raw = nullptr, *d_mod[2] = {nullptr, nullptr};
    uint8_t *d_bits = nullptr; float *d_stats = nullptr; int32_t *d_nfr = nullptr; int64_t *d_cons = nullptr;
    uint8_t *d_status = nullptr, *d_payload = nullptr; int32_t *d_linfo = nullptr;
    HIPOK(hipMalloc(&d_raw, 2 * raw_cap));
    if (D > 1) { HIPOK(hipMalloc(&d_mod[0], bps_mod * mod_cap)); HIPOK(hipMalloc(&d_mod[1], bps_mod * mod_cap)); }
    HIPOK(hipMalloc((void **)&d_bits, (size_t)max_frames * info.Nbits));
    HIPOK(hipMalloc((void **)&d_stats, sizeof(float) * (size_t)max_frames * PIRIP_STATS_PER_FRAME));
    HIPOK(hipMalloc((void **)&d_nfr, sizeof(int32_t)));
    HIPOK(hipMalloc((void **)&d_cons, sizeof(int64_t)));
    if (ldpc) {
        HIPOK(hipMalloc((void **)&d_status, (size_t)max_frames));
        HIPOK(hipMalloc((void **)&d_payload, (size_t)max_frames * li.data_bytes));
        HIPOK(hipMalloc((void **)&d_linfo, sizeof(int32_t) * (size_t)max_frames * PIRIP_LDPC_INFO_PER_CALL));
    }
    std::vector<uint8_t> bits((size_t)max_frames * info.Nbits), status((size_t)max_frames), payload((size_t)max_frames * (ldpc ? li.data_bytes : 1));
    std::vector<int32_t> linfo((size_t)max_frames * PIRIP_LDPC_INFO_PER_CALL);
    std::vector<float> stats((size_t)max_frames * PIRIP_STATS_PER_FRAME), Sf(info.Ndft), timing_acc;
    int cur = 0;
    long total_in = 0, since_json = 0, call_no = 0;
    long frame_periods = 0, period_bits = 0, period_rem = 0;   // -v: bits_per_frame-long periods of demodulator output since start, bits into the
                                                     // current one, and what was left over when it began (upstream's cycling `nbits`)
    double modem_samples = 0.0;                      // modem-rate samples demodulated so far (the sample clock of -L)
    long next_nin = info.N;
    for (;;) {
        size_t want = blk;
        if (raw_have + want > raw_cap) want = raw_cap - raw_have;
        if (nsamples && total_in + (long)want > nsamples) want = (size_t)(nsamples - total_in);
        const size_t got = want ? fread(raw.data() + 2 * raw_have, 2, want, fin) : 0;
        total_in += (long)got; raw_have += got;
        HIPOK(hipMemcpy(d_raw, raw.data(), 2 * raw_have, hipMemcpyHostToDevice));
        const void *d_in = d_raw;
        int64_t n_in = (int64_t)raw_have;
        if (D > 1) {
            // decimate on the device straight behind the carried modem samples: no host hop between decimator and demodulator
            const int64_t nout = pirip_hip_decim_nout(dec, (int64_t)raw_have);
            if (nout > 0) {
                if (mod_have + (size_t)nout > mod_cap) { fprintf(stderr, "rtl_fsk: internal buffer sizing error\n"); return 2; }
                rc = pirip_hip_decim_batch(dec, (const uint8_t *)d_raw, 0, (int64_t)raw_have, (char *)d_mod[cur] + bps_mod * mod_have, 0, 1, nullptr);
                if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: decimator: %s\n", pirip_hip_strerror(rc)); return 2; }
                mod_have += (size_t)nout;
                const size_t used = (size_t)nout * D;             // overlap carry: consumed = D * outputs
                memmove(raw.data(), raw.data() + 2 * used, 2 * (raw_have - used));
                raw_have -= used;
            }
            d_in = d_mod[cur];
            n_in = (int64_t)mod_have;
        }
        // --code: demodulator and FSK_LDPC receiver in one call (bit LLRs handed over on the device); else bits out
        if (ldpc) rc = pirip_hip_fsk_ldpc_rx_batch(h, ldpc, d_in, 0, n_in, d_status, d_payload, d_linfo, d_stats, 0, d_nfr, d_cons, max_frames, nullptr);
        else rc = pirip_hip_demod_batch(h, d_in, 0, n_in, d_bits, 0, nullptr, 0, d_stats, 0, d_nfr, d_cons, max_frames, nullptr);
        if (rc != PIRIP_OK) { fprintf(stderr, "rtl_fsk: %s\n", pirip_hip_strerror(rc)); return 2; }
        HIPOK(hipDeviceSynchronize());
        int32_t nf = 0; int64_t cons = 0;
        HIPOK(hipMemcpy(&nf, d_nfr, sizeof(nf), hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(&cons, d_cons, sizeof(cons), hipMemcpyDeviceToHost));
        if (nf > 0) HIPOK(hipMemcpy(stats.data(), d_stats, sizeof(float) * (size_t)nf * PIRIP_STATS_PER_FRAME, hipMemcpyDeviceToHost));
        if (ldpc && nf > 0) {
            HIPOK(hipMemcpy(status.data(), d_status, (size_t)nf, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(payload.data(), d_payload, (size_t)nf * li.data_bytes, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(linfo.data(), d_linfo, sizeof(int32_t) * (size_t)nf * PIRIP_LDPC_INFO_PER_CALL, hipMemcpyDeviceToHost));
        } else if (!ldpc && nf > 0) {
            HIPOK(hipMemcpy(bits.data(), d_bits, (size_t)nf * info.Nbits, hipMemcpyDeviceToHost));
        }
        // carry the unconsumed modem-rate tail in front of the next block
        if (D > 1) {
            const size_t left = mod_have - (size_t)cons;
            if (left) HIPOK(hipMemcpy(d_mod[cur ^ 1], (char *)d_mod[cur] + bps_mod * (size_t)cons, bps_mod * left, hipMemcpyDeviceToDevice));
            mod_have = left; cur ^= 1;
        } else {
            memmove(raw.data(), raw.data() + 2 * cons, 2 * (raw_have - (size_t)cons));
            raw_have -= (size_t)cons;
        }

        if (!ldpc) fwrite(bits.data(), 1, (size_t)nf * info.Nbits, fout);
        for (int32_t f = 0; f < nf; f++) {
            const float *s = &stats[(size_t)f * PIRIP_STATS_PER_FRAME];
            timing_acc.push_back(s[4]);
            modem_samples += (double)next_nin; next_nin = (long)s[6];     // this call consumed what the previous one announced
            if (ldpc) {
                period_bits += info.Nbits;
                if (period_bits >= li.bits_per_frame) { period_bits -= li.bits_per_frame; period_rem = period_bits; frame_periods++; }
                uint8_t st = status[f];
                uint8_t *pl = &payload[(size_t)f * li.data_bytes];
                const int32_t *in = &linfo[(size_t)f * PIRIP_LDPC_INFO_PER_CALL];
                const bool decoded = in[6] >= 0;
                int ecdd = 0;                                     // --testframes: payload bit errors (bytes 0,1 may carry source / sequence)
                if (testframes && decoded && !tf_bytes.empty())
                    for (int b = 2; b < li.data_bytes - 2; b++) ecdd += __builtin_popcount((unsigned)(pl[b] ^ tf_bytes[(size_t)b]));
                if ((st & PIRIP_RX_BITS) && filter >= 0 && pl[0] == (uint8_t)filter) {     // our own echo: drop (README.md:303-304)
                    st &= (uint8_t)~PIRIP_RX_BITS;
                    memset(pl, 0, (size_t)li.data_bytes);
                }
                if (status_bytes) { if (!(st & PIRIP_RX_BITS)) memset(pl, 0, (size_t)li.data_bytes); fwrite(&st, 1, 1, fout); fwrite(pl, 1, (size_t)li.data_bytes, fout); }
                else if (st & PIRIP_RX_BITS) fwrite(pl, 1, (size_t)li.data_bytes, fout);
                if (verbose && decoded) {
                    char rxst[5] = {(st & PIRIP_RX_BIT_ERRORS) ? 'E' : '-', (st & PIRIP_RX_BITS) ? 'B' : '-', (st & PIRIP_RX_SYNC) ? 'S' : '-',
                                    (st & PIRIP_RX_TRIAL_SYNC) ? 'T' : '-', 0};
                    const double snrdB = 10.0 * log10((double)s[5] * (double)Rs / 3000.0 + 1e-12);
                    const int uw_loc = (int)((in[1] + period_bits) % li.bits_per_frame);      // constant while in sync (see header)
                    fprintf(stderr, "%3ld nbits: %3ld state: %d uw_loc: %3d uw_err: %2d bad_uw: %d snrdB: %4.1f eraw: %3d ecdd: %3d iter: %3d pcc: %3d rxst: %s\n",
                            frame_periods, period_rem, in[0], uw_loc, in[2], in[3], snrdB, in[8], ecdd, in[4], in[5], rxst);
                }
                if (log_frames && (st & PIRIP_RX_BITS)) {
                    const double S = s[8], N = s[9];
                    fprintf(stderr, "%ld Rx frame src: 0x%02x seq: %3d S: %e N: %e SNR: %5.2f dB t_rx: %.4f s\n", (long)time(nullptr), pl[0], pl[1],
                            S, N, 10.0 * log10(S / (N + 1e-30) + 1e-30), modem_samples / (double)Fs);
                }
            } else if (verbose) {
                fprintf(stderr, "%ld nbits: %d snr_lin: %.2f timing: %+.3f f_est: %.0f %.0f\n", call_no, info.Nbits, s[5], s[4], s[0], s[1]);
            }
            call_no++;
        }
        if (fout == stdout || status_bytes) fflush(fout);
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
        if (got < want || (nsamples && total_in >= nsamples) || want == 0) break;
    }
    if (fout != stdout) fclose(fout);
    pirip_hip_destroy(h);
    if (dec) pirip_hip_decim_destroy(dec);
    if (ldpc) pirip_hip_ldpc_destroy(ldpc);
    void *ptrs[] = {d_raw, d_mod[0], d_mod[1], d_bits, d_stats, d_nfr, d_cons, d_status, d_payload, d_linfo};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    return 0;
}
