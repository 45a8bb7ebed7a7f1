"""Every `rtl_fsk` command line of the reference, run VERBATIM through bash in a directory laid out like the reference's
checkout (librtlsdr/build_rtlsdr/src/rtl_fsk, codec2/build_linux/src/fsk_put_test_bits, ~/pirip/codec2/..., src/rtl_fsk,
./rtl_fsk -- symlinks to this repo's binaries), with the dongle replaced by $PIRIP_IQ_FILE and codec2's code name mapped
through $PIRIP_CODE_DIR. Each line's text below is the reference's, character for character, up to the
pipe / redirect that follows it there:
    /root/reference/README.md:114,123,152,172,184,196,239,262,286,292,297
    /root/reference/test/loopback_rtl_fsk.sh:10 (+ test/include.sh:1-10 for its variables)
    /root/reference/script/ping:47, script/frame_repeater:23,36,43
Checked per line: exit status 0, the kernel that served it (wave-per-stream instance or the general kernel), and the byte
stream on stdout against the CPU oracle's whole chain (csdr convert_u8_f -> fir_decimate_cc -> fsk_demod [-> FSK_LDPC rx])
on the same IQ file; the lines the reference pipes into fsk_put_test_bits are piped into it here too."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import sigutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pirip_amd", "bin")
STANDIN = os.path.join(ROOT, "pirip_amd", "data", "standin_256_512_4.code")
RX_SYNC, RX_BITS = 2, 4

# test/include.sh:1-10 (values only)
INCLUDE_SH = dict(dash_host="penetrator", freq="144500000", Fs="240000", Rb="10000", rx_freq=str(144500000 - 2 * 10000),
                  numTxPackets="1000", bitsPerPacket="100", numTxBits=str(1000 * 10000 // 100), passRxPackets="990", rx_secs="15")

# id, cite, the reference's text, what follows it there, shell variables the reference's script defines, "$1",
# modem (Fs after -a, Rs, M, mask spacing, coded, filter address or None, -b), expected RTL rate, kernel expected
LINES = [
    ("readme114", "README.md:114",
     "Fs=240000; tsecs=5; ./librtlsdr/build_rtlsdr/src/rtl_fsk -g 49 -f 144490000 - -n $(($Fs*$tsecs))",
     " | codec2/build_linux/src/fsk_put_test_bits -", {}, None, (240000, 10000, 2, 0, False, None, False), 240000, "wave"),
    ("readme123", "README.md:123",
     "Fs=240000; tsecs=20; ./librtlsdr/build_rtlsdr/src/rtl_fsk -g 1 -f 144490000 - -n $(($Fs*$tsecs)) -u localhost",
     " | codec2/build_linux/src/fsk_put_test_bits -", {}, None, (240000, 10000, 2, 0, False, None, False), 240000, "wave"),
    ("loopback_rtl_fsk", "test/loopback_rtl_fsk.sh:10",
     "rtl-sdr-blog/build_rtlsdr/src/rtl_fsk -g 1 -s $Fs -f $rx_freq - -n $(($Fs*$rx_secs)) -u $dash_host",
     " | codec2/build_linux/src/fsk_put_test_bits -q -p $passRxPackets  -", INCLUDE_SH, None,
     (240000, 10000, 2, 0, False, None, False), 240000, "wave"),
    ("readme152", "README.md:152",
     "./rtl_fsk -w 500E3 -e ff8 -r 1000 -f 144490000 - -u localhost",
     " | ~/pirip/codec2/build_linux/src/fsk_put_test_bits -", {}, None, (240000, 1000, 2, 0, False, None, False), 240000, "block"),
    ("readme172", "README.md:172",
     "./rtl_fsk -s 2400000 -a 80000 -w 500E3 -e ff8 -r 10000 -f 144490000 - -u 192.168.1.100",
     " | ~/pirip/codec2/build_linux/src/fsk_put_test_bits -", {}, None, (80000, 10000, 2, 0, False, None, False), 2400000, "wave"),
    ("readme184", "README.md:184",
     "./src/rtl_fsk -g 49 -f 144490000 - -r 1000 --code  H_256_512_4 -v -u localhost",
     None, {}, None, (240000, 1000, 2, 0, True, None, False), 240000, "block"),
    ("readme196", "README.md:196",
     "./src/rtl_fsk -g 1 -f 144490000 - -a 100000 -r 10000 --code  H_256_512_4 -v -u localhost --testframes",
     None, {}, None, (100000, 10000, 2, 0, True, None, False), 1800000, "wave"),
    ("readme239", "README.md:239",
     "./src/rtl_fsk -g 49 -f 144490000 - -r 1000 -m 4 --code  H_256_512_4 -v -u localhost --testframes --mask 2000 -e 0xfff",
     None, {}, None, (240000, 1000, 4, 2000, True, None, False), 240000, "block"),
    ("readme262", "README.md:262",
     "./src/rtl_fsk -g 49 -f 144490000 - -a 200000 -r 10000 -m 4 --code  H_256_512_4 -v -u localhost --testframes --mask 10000 -e 0xfff",
     None, {}, None, (200000, 10000, 4, 10000, True, None, False), 1800000, "wave"),
    ("readme286", "README.md:286",
     "./src/rtl_fsk -g 30 -f 144490000 - -r 10000 -m 2 -a 180000 --code H_256_512_4 -v -u localhost --testframes -m 4 --mask 10000",
     None, {}, None, (180000, 10000, 4, 10000, True, None, False), 1800000, "wave"),
    ("readme292", "README.md:292",
     "./src/rtl_fsk -g 49 -f 144490000 - -a 200000 -r 10000 --code  H_256_512_4 --mask 10000 --filter 0x2 -q",
     None, {}, None, (200000, 10000, 2, 10000, True, 0x2, False), 1800000, "wave"),
    ("readme297", "README.md:297",
     "./src/rtl_fsk -g 49 -f 144490000 - -a 200000 -r 10000 --code  H_256_512_4 --mask 10000 --filter 0x1",
     None, {}, None, (200000, 10000, 2, 10000, True, 0x1, False), 1800000, "wave"),
    ("ping47", "script/ping:47",
     "rtl_fsk -g ${GAIN} -f 144490000 - -a 40000 -r ${RS} --code  H_256_512_4  -L $1 -u localhost",
     None, dict(GAIN="40", RS="1000"), "--filter 0x1", (40000, 1000, 2, 0, True, 0x1, False), 240000, "wave"),
    ("frame_repeater23", "script/frame_repeater:23",
     "rtl_fsk -g ${GAIN} -f 144490000 - -a 40000 -r ${RS} --code  H_256_512_4  -L $1",
     None, dict(GAIN="40", RS="1000"), "", (40000, 1000, 2, 0, True, None, False), 240000, "wave"),
    ("frame_repeater36", "script/frame_repeater:36",
     "rtl_fsk -g ${GAIN} -f 144490000 - -a 40000 -r ${RS} --code ${CODE} --filter ${TERM_ADDR} -q -b",
     None, dict(GAIN="40", RS="1000", CODE="H_256_512_4", TERM_ADDR="0x2"), None, (40000, 1000, 2, 0, True, 0x2, True), 240000, "wave"),
    ("frame_repeater43", "script/frame_repeater:43",
     "rtl_fsk -g ${GAIN} -f 144490000 - -a 40000 -r ${RS} --code ${CODE} --filter ${TERM_ADDR} -q -b -v",
     None, dict(GAIN="40", RS="1000", CODE="H_256_512_4", TERM_ADDR="0x2"), None, (40000, 1000, 2, 0, True, 0x2, True), 240000, "wave"),
]


def _rtl_fsk_P(Ts):
    P = Ts
    while P > 10 and P % 2 == 0:
        P //= 2
    return P if P >= 4 else Ts


def _interp(x, D):
    """x D linear interpolation of complex float [n,2] (tlininterp of the reference's bench Tx chain, README.md:142)."""
    if D == 1:
        return x
    n = x.shape[0]
    t = np.arange((n - 1) * D) / float(D)
    i0 = np.floor(t).astype(np.int64)
    fr = (t - i0)[:, None].astype(np.float32)
    return (1 - fr) * x[i0] + fr * x[i0 + 1]


def _layout(tmp):
    """the reference's on-disk paths (build_codec2.sh / build_rtlsdr.sh), as symlinks to this repo's tools"""
    links = {
        "librtlsdr/build_rtlsdr/src/rtl_fsk": "rtl_fsk", "rtl-sdr-blog/build_rtlsdr/src/rtl_fsk": "rtl_fsk",
        "src/rtl_fsk": "rtl_fsk", "rtl_fsk": "rtl_fsk", "path/rtl_fsk": "rtl_fsk",
        "codec2/build_linux/src/fsk_put_test_bits": "fsk_put_test_bits",
        "home/pirip/codec2/build_linux/src/fsk_put_test_bits": "fsk_put_test_bits",
        "codes/H_256_512_4.code": None,
    }
    for rel, tool in links.items():
        dst = tmp / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        os.symlink(STANDIN if tool is None else os.path.join(BIN, tool), dst)


def _signal(oracle, modem, rtlFs, seed):
    """u8 IQ at the RTL rate for one command line, and what was sent. Tones sit where the reference puts them
    (README.md:142,257: tone1 = Rs ... 10 kHz above the tuned frequency, shift = Rs or 2 Rs)."""
    Fs, Rs, M, mask, coded, _, _ = modem
    D = rtlFs // Fs
    rng = np.random.default_rng(seed)
    spacing = mask if mask else (2000 if Rs == 1000 else 10000)
    f1 = 2 * Rs if Rs == 1000 else 10000
    cfg = dict(Fs=Fs, Rs=Rs, M=M, P=_rtl_fsk_P(Fs // Rs), f1=f1, shift=spacing)
    Ts = Fs // Rs
    if coded:
        def framer(src, n):
            p = subprocess.run([os.path.join(BIN, "fsk_ldpc_framer"), "--code", STANDIN, "-m", str(M), "--testframes", str(n), "--seq",
                                "--source", hex(src), "/dev/zero", "-"], capture_output=True)
            assert p.returncode == 0, p.stderr
            return np.frombuffer(p.stdout, dtype=np.uint8)
        bursts = [framer(0x1, 2), framer(0x2, 2)]
        gap = 30 * Ts
        segs = [np.zeros((gap + int(rng.integers(0, Ts)), 2), dtype=np.float32)]
        for b in bursts:
            segs.append(sigutil.mod_complex(oracle, cfg, b))
            segs.append(np.zeros((3 * gap, 2), dtype=np.float32))
        segs.append(np.zeros((600 * Ts, 2), dtype=np.float32))       # a frame is decoded once a further frame's worth of bits has arrived
        x = np.concatenate(segs)
        sent = bursts
    else:
        nbits = 100000 + 2000 if modem[1] == 10000 and Fs == 240000 and seed == 2 else 6000
        bits = oracle.get_test_bits(nbits)
        # the receiver is started before the transmitter (test/loopback_rtl_fsk.sh:10-17): a noise-only lead-in, then the signal
        x = np.concatenate([np.zeros((10 * Ts + int(rng.integers(0, Ts)), 2), dtype=np.float32), sigutil.mod_complex(oracle, cfg, bits)])
        sent = bits
    eb = 4.0 * Ts / np.log2(M)
    # Eb/N0 at the modem rate: 12 dB for the coded lines; the uncoded ones are piped into fsk_put_test_bits, whose PASS needs a
    # bit error rate of 0 unless -b is given [UPSTREAM-RECALLED default] -- the reference runs them over a cable with a 60 dB
    # attenuator (README.md:126), i.e. error free: 25 dB here
    ebno = 12.0 if coded else 25.0
    sigma = np.sqrt(eb / (10 ** (ebno / 10.0)) / 2.0)
    x = (x + rng.normal(0.0, sigma, x.shape)).astype(np.float32)
    return oracle.quantise_cu8(_interp(x, D), amp=20.0), cfg, sent


def _oracle_chain(oracle, u8, modem, rtlFs, cfg):
    """csdr convert_u8_f [-> fir_decimate_cc D] -> fsk_demod [-> FSK_LDPC rx] on the CPU: the bytes rtl_fsk must print"""
    Fs, Rs, M, mask, coded, filt_addr, status_bytes = modem
    D = rtlFs // Fs
    L = oracle.lib()
    o = oracle.OracleFsk(Fs, Rs, M, P=cfg["P"], tone_spacing=mask if mask else 100, est_min=Rs // 2, est_max=Fs // 2, mask=bool(mask))
    if D > 1:
        f = np.zeros(u8.shape, dtype=np.float32)
        L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
        ntaps = L.oracle_firdes_filter_len(0.05)
        tp = np.zeros(80, dtype=np.float32)
        L.oracle_firdes_lowpass_f_hamming(tp.ctypes.data, ntaps, 0.5 / D)
        y = np.zeros((u8.shape[0] // D + 1, 2), dtype=np.float32)
        n_out = L.oracle_fir_decimate_cc(f.ctypes.data, y.ctypes.data, u8.shape[0], D, tp.ctypes.data, 80)
        r = o.demod(y[:n_out], oracle.IN_CF32)
    else:
        r = o.demod(u8, oracle.IN_CU8_CSDR)
    if not coded:
        return r["bits"].reshape(-1), o.Nbits
    code = oracle.parse_code_file(STANDIN)
    st, pl, _ = oracle.OracleLdpc(code, M).rx(r["rx_filt"])
    st = st.copy(); pl = pl.copy()
    good = (st & RX_BITS) != 0
    if filt_addr is not None:
        own = good & (pl[:, 0] == filt_addr)
        st[own] &= ~np.uint8(RX_BITS)
        pl[own] = 0
        good = (st & RX_BITS) != 0
    pl[~good] = 0
    if status_bytes:
        return np.concatenate([st[:, None], pl], axis=1).reshape(-1), 33
    return pl[good].reshape(-1), 32


@pytest.mark.gpu
@pytest.mark.parametrize("line", LINES, ids=[ln[0] for ln in LINES])
def test_reference_rtl_fsk_command_line_verbatim(oracle, built_lib, tmp_path, line):
    name, cite, text, tail, shvars, arg1, modem, rtlFs, kernel = line
    _layout(tmp_path)
    seed = 2 if name == "loopback_rtl_fsk" else 3 + zlib.crc32(name.encode()) % 1000
    u8, cfg, sent = _signal(oracle, modem, rtlFs, seed)
    iq = tmp_path / "capture.iq8"
    u8.tofile(iq)
    env = dict(os.environ, PIRIP_IQ_FILE=str(iq), PIRIP_CODE_DIR=str(tmp_path / "codes"), PIRIP_RTL_FSK_BANNER="1",
               HOME=str(tmp_path / "home"), PATH=str(tmp_path / "path") + ":" + os.environ["PATH"], **shvars)
    args = ["bash", "-c", text, "bash"] + ([arg1] if arg1 is not None else [])
    p = subprocess.run(args, cwd=tmp_path, env=env, capture_output=True, timeout=600)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, (cite, err[-2000:])
    banner = [ln for ln in err.split("\n") if ln.startswith("rtl_fsk: rtl rate")]
    assert len(banner) == 1, err[-2000:]
    assert f"rtl rate {rtlFs} " in banner[0] and f" Fs {modem[0]} Rs {modem[1]} M {modem[2]} " in banner[0], banner[0]
    assert f"kernel {kernel}" in banner[0], (cite, banner[0])
    want, rec = _oracle_chain(oracle, u8, modem, rtlFs, cfg)
    got = np.frombuffer(p.stdout, dtype=np.uint8)
    # the block-wise reader may hold back the last frame of the file (a partial decimator block is dropped, like csdr's)
    n = (min(got.size, want.size) // rec) * rec
    assert got.size <= want.size and got.size >= want.size - 2 * rec, (cite, got.size, want.size)
    assert np.array_equal(got[:n], want[:n]), (cite, np.where(got[:n] != want[:n])[0][:10])
    coded = modem[4]
    if not coded:
        res = oracle.put_test_bits(got)
        assert res["errors"] <= res["bits"] * 2e-3 and res["packets"] >= 50, res
        # the consumer the reference pipes into
        p2 = subprocess.run(["bash", "-c", text + tail, "bash"], cwd=tmp_path, env=env, capture_output=True, timeout=600)
        # fsk_put_test_bits' verdict is a property of the bits (PASS = enough packets and a bit error rate of 0 unless -b is
        # given [UPSTREAM-RECALLED]): it must be what the same rule says about the oracle's bits
        verdict = res["errors"] == 0 and (name != "loopback_rtl_fsk" or res["packets"] >= 990)
        assert p2.returncode == (0 if verdict else 1), (cite, p2.stderr[-2000:])
        assert (b"PASS" if verdict else b"FAIL") in p2.stderr + p2.stdout
        if name == "loopback_rtl_fsk":
            assert verdict, res                                                    # the hardware-in-the-loop line must pass
    else:
        filt_addr, status_bytes = modem[5], modem[6]
        if status_bytes:
            r = got.reshape(-1, 33)
            pay = r[(r[:, 0] & RX_BITS) != 0, 1:]
        else:
            pay = got.reshape(-1, 32)
        srcs = sorted(set(int(b) for b in pay[:, 0]))
        assert srcs == [a for a in (1, 2) if a != filt_addr], (cite, srcs)        # both bursts arrive, our own address is dropped
        assert pay.shape[0] >= (3 if filt_addr is None else 1)
        tf = np.packbits(sent[0][sent[0].size - 2 * 544 + 32:][:256])         # (preamble | UW + 256 data + 256 parity) x 2
        assert all(np.array_equal(f[2:30], tf[2:30]) for f in pay)                # 0 coded errors
        if " -v" in text:
            lines = [ln for ln in err.split("\n") if "rxst:" in ln]
            assert len(lines) >= 3
            # first column: consecutive frame-period counter; nbits cycles by (Nbits - bits_per_frame % Nbits) per frame while
            # in sync (README.md:200-208: +6 at 50 bits per call; :241-243: +56 at 100); uw_loc constant within a burst
            Nbits = 50 * (modem[2] // 2)
            cnt = [int(ln.split()[0]) for ln in lines]
            nb = [int(ln.split("nbits:")[1].split()[0]) for ln in lines]
            loc = [int(ln.split("uw_loc:")[1].split()[0]) for ln in lines]
            step = (Nbits - 544 % Nbits) % Nbits
            assert all(0 <= x < Nbits for x in nb), nb
            pairs = [(i, i + 1) for i in range(len(lines) - 1) if cnt[i + 1] == cnt[i] + 1 and loc[i] == loc[i + 1]]   # same burst
            assert pairs, (cnt, loc)
            for i, j in pairs:
                assert (nb[j] - nb[i]) % Nbits == step, (lines[i], lines[j])
            if "--testframes" in text:
                assert all(int(ln.split("ecdd:")[1].split()[0]) == 0 for ln in lines)
        if " -L" in text:
            logs = [ln for ln in err.split("\n") if " Rx frame src:" in ln]
            assert len(logs) == pay.shape[0], (len(logs), pay.shape[0])
            for ln, f in zip(logs, pay):
                assert int(ln.split()[0]) > 1_600_000_000                         # wall-clock arrival time, like `date +%s` in script/ping:30
                assert int(ln.split("src:")[1].split()[0], 16) == f[0] and int(ln.split("seq:")[1].split()[0]) == f[1]
                S = float(ln.split(" S:")[1].split()[0]); N = float(ln.split(" N:")[1].split()[0]); snr = float(ln.split("SNR:")[1].split()[0])
                assert S > N > 0 and abs(snr - 10 * np.log10(S / N)) < 0.02 and 5.0 < snr < 25.0, ln
            t = [float(ln.split("t_rx:")[1].split()[0]) for ln in logs]
            assert all(b > a for a, b in zip(t, t[1:])) and t[-1] < u8.shape[0] / rtlFs + 1e-6


@pytest.mark.gpu
def test_rtl_fsk_recalled_rules_are_data(oracle, built_lib, tmp_path):
    """The two rules of upstream's rtl_fsk.c this tool holds from recall -- the timing-oversample reduction and the RTL rate when -s is
    absent -- are one struct with today's values as defaults, flipped without a rebuild through PIRIP_RTL_FSK_RULES (the tool-level part
    of the pin-day drill): the banner shows the P and the rate each setting selects, a key that does not exist is refused."""
    iq = tmp_path / "x.iq8"
    np.full(240000 * 2, 127, dtype=np.uint8).tofile(iq)
    exe = os.path.join(ROOT, "pirip_amd", "bin", "rtl_fsk")

    def banner(rules, *argv):
        env = dict(os.environ, PIRIP_IQ_FILE=str(iq), PIRIP_RTL_FSK_BANNER="1")
        if rules:
            env["PIRIP_RTL_FSK_RULES"] = rules
        p = subprocess.run([exe, "-q", *argv, "-"], env=env, capture_output=True, timeout=300)
        ln = [x for x in p.stderr.decode(errors="replace").split("\n") if x.startswith("rtl_fsk: rtl rate")]
        return p.returncode, (ln[0] if ln else p.stderr.decode(errors="replace"))
    rc, b = banner(None)
    assert rc == 0 and "rtl rate 240000 Fs 240000 Rs 10000 M 2 P 6 " in b, b              # Ts = 24: halved twice (24 -> 12 -> 6)
    rc, b = banner("p_rule=1")
    assert rc == 0 and " P 24 " in b, b
    rc, b = banner("p_rule=2")
    assert rc == 0 and " P 8 " in b, b
    rc, b = banner("p_max=12")
    assert rc == 0 and " P 12 " in b, b
    rc, b = banner("default_rate=1200000", "-a", "240000")
    assert rc == 0 and "rtl rate 1200000 Fs 240000 " in b and "decimation 5 " in b, b
    rc, b = banner("no_such_rule=1")
    assert rc == 2 and "PIRIP_RTL_FSK_RULES" in b, b
