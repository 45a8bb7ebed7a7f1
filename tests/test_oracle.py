"""CPU tests of the oracle (test infrastructure) against what can pin it:
committed fixtures, numpy's FFT, non-coherent FSK theory, and self-consistent loopback.
The reference has no golden vectors for this path -> "parity unpinned"."""
import os

import numpy as np
import pytest

import sigutil

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rx(ob, c, **kw):
    return ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], **kw)


def test_test_frame_is_the_recalled_glibc_sequence(oracle):
    # srand(158324); rand()&1 under glibc gives this 100-bit frame
    want = "0101111110001000101010010010010100011000011111001110111111100111110011100011110001110001111000110101"
    got = "".join(str(int(b)) for b in oracle.get_test_bits(100))
    assert got == want
    g = np.load(os.path.join(GOLD, "cfg1_clean.npz"))
    assert np.array_equal(g["test_frame"], oracle.get_test_bits(100))


def test_golden_cfg1_clean(oracle):
    g = np.load(os.path.join(GOLD, "cfg1_clean.npz"))
    r = _rx(oracle, sigutil.CFG1).demod(g["iq_u8"], oracle.IN_CU8_FSKDEMOD)
    assert np.array_equal(r["bits"], g["bits"])
    assert np.array_equal(r["rx_filt"], g["rx_filt"])          # same machine class: bit-identical
    assert np.array_equal(r["stats"][:, :4], g["stats"][:, :4])
    # decoded bits are the transmitted test frames (after the demod's start-up delay)
    res = oracle.put_test_bits(r["bits"])
    assert res["errors"] == 0 and res["packets"] >= 4


def test_golden_cfg1_noisy_and_cfg4(oracle):
    g = np.load(os.path.join(GOLD, "cfg1_noisy8dB.npz"))
    r = _rx(oracle, sigutil.CFG1).demod(g["iq_u8"], oracle.IN_CU8_FSKDEMOD)
    assert np.array_equal(r["bits"], g["bits"])
    np.testing.assert_allclose(r["rx_filt"], g["rx_filt"], rtol=0, atol=1e-6 * np.abs(g["rx_filt"]).max())
    g = np.load(os.path.join(GOLD, "cfg4_clean.npz"))
    r = _rx(oracle, sigutil.CFG4).demod(g["iq_u8"], oracle.IN_CU8_FSKDEMOD)
    assert np.array_equal(r["bits"], g["bits"])
    # 4-FSK: the decoded stream contains the transmitted bits exactly (alignment found by search)
    tx = g["tx_bits"]; rxb = r["bits"].reshape(-1)
    ok = any(np.array_equal(rxb[o:o + 400], tx[s:s + 400]) for o in range(0, 200, 2) for s in range(0, 200, 2))
    assert ok


def test_kiss_fft_restatement_matches_numpy_and_fixture(oracle):
    import ctypes as C
    L = oracle.lib()
    L.kiss_fft_oracle_alloc.restype = C.c_void_p
    L.kiss_fft_oracle_alloc.argtypes = [C.c_int, C.c_int]
    L.kiss_fft_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    g = np.load(os.path.join(GOLD, "kiss_fft.npz"))
    for n in (256, 512):
        x = g[f"x{n}"]
        y = np.zeros_like(x)
        L.kiss_fft_oracle(L.kiss_fft_oracle_alloc(n, 0), x.ctypes.data, y.ctypes.data)
        assert np.array_equal(y, g[f"y{n}"])
        ref = np.fft.fft(x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64))
        got = y[:, 0].astype(np.float64) + 1j * y[:, 1]
        assert np.max(np.abs(got - ref)) < 2e-5 * np.max(np.abs(ref))


def test_loopback_cfg1_600k_bits_zero_errors(oracle):
    """BASELINE config 1: fsk_get_test_bits 600000 | fsk_mod | u8 | fsk_demod -d -p 24 | fsk_put_test_bits."""
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 600000)
    r = _rx(oracle, c).demod(u8, oracle.IN_CU8_FSKDEMOD, want_filt=False)
    assert r["nframes"] == 12000
    res = oracle.put_test_bits(r["bits"], packet_pass=5990)
    assert res["errors"] == 0 and res["pass"], res
    # tone estimates are bin-quantised (937.5 Hz) around 10 and 20 kHz
    assert set(np.unique(r["stats"][50:, 0])) <= {9375.0, 10312.5} and set(np.unique(r["stats"][50:, 1])) <= {19687.5, 20625.0}


@pytest.mark.parametrize("ebno_db", [6.0, 9.0])
def test_ber_against_noncoherent_fsk_theory(oracle, ebno_db):
    """Independent anchor: BER of non-coherent 2-FSK is 0.5*exp(-Eb/2N0)."""
    c = sigutil.CFG1
    nbits = 200000
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = sigutil.add_awgn(sigutil.mod_complex(oracle, c, bits), ebno_db, c, rng)
    r = _rx(oracle, c).demod(x, oracle.IN_CF32, want_filt=False)
    rxb = r["bits"].reshape(-1)
    # align: the demod delays by a whole number of symbols < 2 frames
    best = min(((np.mean(rxb[d:d + 150000] != bits[:150000]), d) for d in range(0, 120)), key=lambda t: t[0])
    ber, _ = best
    theory = 0.5 * np.exp(-(10 ** (ebno_db / 10)) / 2)
    assert theory * 0.7 < ber < theory * 2.0, (ber, theory)   # implementation loss < ~1 dB, MC error small


def test_nin_feedback_tracks_sample_clock_offset(oracle):
    """Timing drift (resampled stream) must trigger nin = N +- Ts/4 and keep 0 errors."""
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(60000))
    n = x.shape[0]
    # +300 ppm sample clock: linear-interpolated resample
    t = np.arange(int(n / 1.0003)) * 1.0003
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    y = (1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, n - 1)]
    r = _rx(oracle, c).demod(oracle.quantise_cu8(y), oracle.IN_CU8_FSKDEMOD, want_filt=False)
    nins = r["stats"][:, 6]
    assert (nins != 1200).any()
    assert oracle.put_test_bits(r["bits"])["errors"] == 0


def test_csdr_stage_fixture_and_tap_design(oracle):
    L = oracle.lib()
    g = np.load(os.path.join(GOLD, "csdr_decim45.npz"))
    n = L.oracle_firdes_filter_len(0.05)
    assert n == 79            # int(4.0/0.05f) = 79 (already odd); csdr pads to 80 for its NEON path
    taps = np.zeros(n, dtype=np.float32)
    L.oracle_firdes_lowpass_f_hamming(taps.ctypes.data, n, 0.5 / 45)
    assert np.array_equal(taps, g["taps"])
    assert abs(float(taps.sum()) - 1.0) < 1e-6 and np.allclose(taps, taps[::-1])
    u8 = g["iq_u8"]
    f = np.zeros(u8.shape, dtype=np.float32)
    L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
    assert f.min() >= -1.0 and f.max() <= 1.0 and f[u8 == 0].max(initial=-1.0) == -1.0
    # direct evaluation of y[k] = sum h[t] x[45k+t] in float64 agrees to float32 rounding
    yk = np.array([(f[45 * k:45 * k + n].astype(np.float64) * taps[:, None]).sum(0) for k in range(g["y_f32"].shape[0])])
    assert np.max(np.abs(yk - g["y_f32"])) < 1e-6
    assert np.array_equal(g["y_s16"], np.trunc(g["y_f32"] * np.float32(32767)).astype(np.int16))


def test_csdr_stream_block_structure(oracle):
    """csdr's block loop: 16384-sample blocks, (16384-80)//45+1 = 363 outputs and 16335 consumed per block."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    nsamp = 16384 + 16335 * 2 + 1000
    x = rng.standard_normal((nsamp, 2)).astype(np.float32)
    out = np.zeros((4096, 2), dtype=np.float32)
    n = L.oracle_csdr_fir_decimate_stream(x.ctypes.data, nsamp, out.ctypes.data, 4096, 45, 0.05, 16384)
    assert n == 3 * 363
    # continuity across blocks: output k is always the FIR at input offset 45k
    ntaps = L.oracle_firdes_filter_len(0.05)
    taps = np.zeros(ntaps, dtype=np.float32)
    L.oracle_firdes_lowpass_f_hamming(taps.ctypes.data, ntaps, 0.5 / 45)
    for k in (0, 362, 363, 364, 1088):
        ref = (x[45 * k:45 * k + ntaps].astype(np.float64) * taps[:, None]).sum(0)
        assert np.max(np.abs(ref - out[k])) < 1e-5


def test_csdr_taps_against_scipy_firwin(oracle):
    """Independent anchor for the decimator design: csdr's firdes_lowpass_f (windowed sinc, Hamming,
    normalised to unit DC gain) is the textbook design scipy.signal.firwin implements."""
    from scipy.signal import firwin
    L = oracle.lib()
    n = L.oracle_firdes_filter_len(0.05)
    taps = np.zeros(n, dtype=np.float32)
    for D in (45, 50, 5):
        L.oracle_firdes_lowpass_f_hamming(taps.ctypes.data, n, 0.5 / D)
        ref = firwin(n, 1.0 / D, window="hamming")          # cutoff as a fraction of Nyquist = 2*fc
        assert np.max(np.abs(taps - ref)) < 2e-6, D


def test_timing_estimate_tracks_a_known_sample_offset(oracle):
    """Independent anchor for the fine-timing estimator: dropping k leading samples moves
    norm_rx_timing by -k/Ts (mod 1), and the decoded bits stay the transmitted ones."""
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(6000))
    base = None
    for k in (0, 3, 7, 10):
        r = _rx(oracle, c).demod(oracle.quantise_cu8(x[k:]), oracle.IN_CU8_FSKDEMOD, want_filt=False)
        t = float(np.median(r["stats"][20:, 4]))
        # frames re-align by +-Ts/4 steps through nin, so compare modulo a quarter symbol
        if base is None:
            base = t
        d = ((t - base) + k / 24.0 + 0.125) % 0.25 - 0.125
        assert abs(d) < 0.01, (k, t, base)
        assert oracle.put_test_bits(r["bits"])["errors"] == 0


def test_4fsk_ber_against_noncoherent_theory(oracle):
    """4-FSK anchor: symbol error rate of non-coherent orthogonal M-FSK,
    Ps = sum_{k=1}^{M-1} (-1)^{k+1} C(M-1,k)/(k+1) exp(-k/(k+1) Es/N0), bit error rate = Ps * (M/2)/(M-1)."""
    from math import comb, exp
    c = sigutil.CFG4
    ebno_db = 8.0
    rng = np.random.default_rng(7)
    nbits = 200000
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = sigutil.add_awgn(sigutil.mod_complex(oracle, c, bits), ebno_db, c, rng)
    r = _rx(oracle, c).demod(x, oracle.IN_CF32, want_filt=False)
    rxb = r["bits"].reshape(-1)
    ber = min(np.mean(rxb[d:d + 150000] != bits[:150000]) for d in range(0, 240, 2))
    esn0 = 2 * 10 ** (ebno_db / 10)
    ps = sum((-1) ** (k + 1) * comb(3, k) / (k + 1) * exp(-k / (k + 1) * esn0) for k in range(1, 4))
    theory = ps * 2 / 3
    assert theory * 0.6 < ber < theory * 2.5, (ber, theory)


def test_eye_diagram_of_a_clean_signal(oracle):
    """MODEM_STATS.rx_eye as restated (oracle_fsk_get_eye): 8 / M traces per tone of two symbols each, row = trace * M + tone.
    On a clean signal sampled at the symbol centre exactly one tone's trace is near its peak and the others near zero, the
    normalised maximum is 1, and the un-normalised peak is the frame's largest soft-decision magnitude (the same integrator outputs)."""
    for c in (sigutil.CFG1, sigutil.CFG4):
        M, P = c["M"], c["P"]
        u8, _ = sigutil.make_u8_stream(oracle, c, 3000, amp=32.0)
        o = _rx(oracle, c)
        r = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
        assert r["nframes"] > 5
        eye, raw = o.eye(), o.eye(normalise=False)
        assert eye.shape == raw.shape == ((8 // M) * M, 2 * P) and eye.max() == 1.0
        assert raw.max() == pytest.approx(float(r["rx_filt"][-1].max()), rel=0.03)
        # at every trace's best-aligned position the tones' magnitudes are one-hot
        tr = eye.reshape(8 // M, M, 2 * P)
        for t in tr:
            j = int(np.argmax(t.max(axis=0)))
            col = np.sort(t[:, j])
            assert col[-1] > 0.9 and col[-2] < 0.25


def test_pin_day_drill_names_the_recalled_constant_that_differs(oracle):
    """oracle/pin_against_ref.py's repair search, exercised without upstream: let "upstream" be this restatement's own command-line tool
    with ONE recalled constant at its other value (PIRIP_RECALLED); the search over single-field flips must name exactly that field.
    (Noisy input: on clean signals several of the constants cannot be seen in the bits at all.)"""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("pin_against_ref", os.path.join(ROOT, "oracle", "pin_against_ref.py"))
    pin = importlib.util.module_from_spec(spec); spec.loader.exec_module(pin)
    orc = os.path.join(ROOT, "oracle", "build", "fsk_demod_oracle")
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 20000, seed=3, ebno_db=7.0, random_bits=True, amp=18.0)
    args = ["--fsk_lower", "500", "--fsk_upper", "25000", "-d", "-p", "24", "2", "240000", "10000"]
    base = subprocess.run([orc] + args + ["-", "-"], input=u8.tobytes(), capture_output=True, check=True).stdout
    for field in ("nin_threshold", "nin_step_div", "tc", "ndft_rule"):
        alt = oracle.RECALLED_ALTERNATIVES[field]
        fake_upstream = pin.run_env(orc, args + ["-", "-"], u8.tobytes(), {"PIRIP_RECALLED": f"{field}={alt}"})
        assert fake_upstream and fake_upstream != base, field
        rep = pin.which_field(orc, args, u8.tobytes(), fake_upstream, oracle.RECALLED_ALTERNATIVES)
        assert rep["flip_repairs"] == [f"{field}={alt}"], (field, rep["flip_repairs"], rep["furthest"])
    assert subprocess.run([orc] + args + ["-", "-"], input=b"", capture_output=True, env=dict(os.environ, PIRIP_RECALLED="no_such_field=1")).returncode == 2
