"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical
seeded inputs. Bar: decoded bits, tone estimates (bin-quantised), the nin sequence
and the smoothed spectrum Sf are bit-exact; soft magnitudes rx_filt within RX_FILT_TOL of the
frame-set peak (stated float tolerance: the down-conversion oscillator is not the upstream
recursion and window sums are evaluated in a different order)."""
import os
import subprocess

import numpy as np
import pytest

import sigutil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pirip_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")
RX_FILT_TOL = 1e-4        # relative to the peak magnitude of the compared block
SNR_TOL = 2e-3            # relative, SNRest (ratio of two reductions)
TIMING_TOL = 5e-5         # absolute, norm_rx_timing in symbols: the oracle's recursive timing phasor drifts ~1e-5 over 1224 steps


def _pair(ob, c, fmt_o, fmt_h, nstreams=1, mask=0):
    import pirip_amd
    o = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"],
                     tone_spacing=mask if mask else 100, mask=bool(mask))
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"],
                           mask=mask, in_format=fmt_h, nstreams=nstreams)
    return o, h


def _compare(ro, rh, tol=RX_FILT_TOL, allow_near_tie_flips=False, M=2):
    """Exact: frame count, consumed samples, tone estimates, nin sequence, bits.
    Tolerance: rx_filt, norm_rx_timing, SNRest. With allow_near_tie_flips (noisy inputs only) a
    differing bit is accepted -- and counted, the caller prints it -- only where the ORACLE's own
    decision margin (largest minus second largest tone magnitude of that symbol, any M) is below
    2*tol of the peak, i.e. where the two float32 evaluation orders straddle a tie; anything else is
    a failure. (tools/scale_check.py applies the same rule to 10^8 bits; tests/test_scale_check.py.)"""
    assert rh["nframes"] == ro["nframes"] and rh["consumed"] == ro["consumed"]
    assert np.array_equal(rh["stats"][:, :4], ro["stats"][:, :4]), "tone estimates differ"
    assert np.array_equal(rh["stats"][:, 6], ro["stats"][:, 6]), "nin sequence differs"
    nflips = 0
    if not np.array_equal(rh["bits"], ro["bits"]):
        diff = np.argwhere(rh["bits"] != ro["bits"])
        assert allow_near_tie_flips, f"{len(diff)} bit differences"
        filt = ro["rx_filt"]; peak = float(np.abs(filt).max())
        nbits = rh["bits"].shape[1]
        bps = 1 if M == 2 else 2
        nsym = nbits // bps
        assert filt.shape[1] == M * nsym
        for fr, b in diff:
            mags = np.sort(filt[fr].reshape(M, nsym)[:, b // bps])
            margin = float(mags[-1] - mags[-2]) / peak
            assert margin < 2 * tol, f"bit flip at frame {fr} bit {b} with margin {margin:.2e} of peak"
        nflips = len(diff)
    # The fine-timing estimate is the angle of a sum of (Nsym+1)*P terms; under noise that sum nearly cancels in a few frames per
    # 10^4 (tools/scale_check.py counts them: "ill-conditioned"), the two summation orders then give angles more than TIMING_TOL apart
    # and every interpolated magnitude of such a frame moves with the angle. Noisy tests accept a handful of those frames at a
    # looser magnitude tolerance; everywhere else -- and in every frame of a noise-free test -- both tolerances hold as stated.
    good = np.ones(ro["nframes"], dtype=bool)
    if ro["nframes"]:
        dt = np.abs(rh["stats"][:, 4] - ro["stats"][:, 4])                          # norm_rx_timing
        # (frames longer than 2400 samples: the caller scales tol by N / 2400, and the timing estimate -- the angle of a sum of the same
        #  correlator outputs -- moves with it: tools/fuzz_parity.py found 5.07e-5 on a noise-free Ts = 100, P = 4 stream, general kernel)
        ttol = TIMING_TOL * max(1.0, tol / RX_FILT_TOL)
        good = dt < ttol
        if allow_near_tie_flips:
            assert (~good).sum() <= max(2, ro["nframes"] // 300) and dt.max() < 100 * ttol, ((~good).sum(), dt.max())
            # ... and a frame is only excused when the ORACLE's own phasor sum is ill-conditioned: |t_c| / sum|terms| -- the
            # binding's "timing_cond", median 0.02-0.03 on these signals -- below 0.005, where a 1e-6 relative difference of the terms
            # moves the angle by more than TIMING_TOL / (2 pi 0.005) ~ 3e-5 symbols; a badly wrong frame with a healthy sum fails here
            if "timing_cond" in ro and (~good).any():
                assert np.all(ro["timing_cond"][~good] < 0.005 * max(1.0, tol / RX_FILT_TOL)), (ro["timing_cond"][~good], dt[~good])
        else:
            assert good.all(), dt.max()
    if ro["rx_filt"] is not None and rh["rx_filt"] is not None and ro["nframes"]:
        peak = max(float(np.max(np.abs(ro["rx_filt"]))), 1e-30)
        err = np.abs(rh["rx_filt"].astype(np.float64) - ro["rx_filt"].astype(np.float64)).max(axis=1) / peak
        # a frame's magnitudes are interpolated at its timing estimate: two estimates dt symbols apart (dt <= TIMING_TOL in a good frame) move
        # them by up to ~3 dt of the peak (the matched filter's slope), on top of the correlator tolerance -- found by tools/fuzz_parity.py:
        # one noisy frame in 1.6 x 10^4 noisy draws with dt = 4.0e-5 and an error of 1.013e-4
        assert np.all(err[good] < tol + 3.0 * dt[good]), float((err[good] - 3.0 * dt[good]).max())
        assert err[~good].max(initial=0.0) < 100 * tol, err[~good].max()
        sn_o, sn_h = ro["stats"][:, 5].astype(np.float64), rh["stats"][:, 5].astype(np.float64)
        rel = np.abs(sn_h - sn_o) / np.maximum(sn_o, 1e-9)
        inv = np.abs(1.0 / np.maximum(sn_h, 1e-9) - 1.0 / np.maximum(sn_o, 1e-9))
        # SNRest = sig/nse: on clean signals nse is ~1e-3 of sig, so compare the noise fraction
        assert np.all(((rel < SNR_TOL) | (inv < 5e-5))[good]), (rel.max(), inv.max())
        # rx_sig_pow / rx_nse_pow (what rtl_fsk -L logs as S and N): sums of Nsym terms in a different order
        so, sh = ro["stats"][:, 8].astype(np.float64), rh["stats"][:, 8].astype(np.float64)
        no, nh = ro["stats"][:, 9].astype(np.float64), rh["stats"][:, 9].astype(np.float64)
        assert np.all((np.abs(sh - so) <= 2 * tol * np.maximum(so, 1e-30))[good]) and np.all((np.abs(nh - no) <= 2 * tol * np.maximum(so, 1e-30))[good]), \
            (np.abs(sh / np.maximum(so, 1e-30) - 1).max(), (np.abs(nh - no) / np.maximum(so, 1e-30)).max())
    return nflips


@pytest.fixture(params=["auto", "general"])
def kernel_choice(request, monkeypatch):
    """Run a test once on the kernel the library picks (the specialised fast kernel for the
    headline configuration) and once with the general kernel forced."""
    if request.param == "general":
        monkeypatch.setenv("PIRIP_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("PIRIP_FORCE_GENERAL", raising=False)
    return request.param


def test_estimator_sqrt_is_correctly_rounded_on_this_device(built_lib):
    """The wave kernel's fast path replaces sqrtf by rsq + one residual step (one transcendental + 5 VALU instead of
    v_sqrt + 8): correct rounding of that sequence is a measured property of the device, so it is measured here --
    x = 0 and every float in [2^-96, FLT_MAX], both device variants, against (float)sqrt((double)x)."""
    import pirip_amd
    m = pirip_amd.selftest_sqrt()
    assert (m & 0xffffffff, m >> 32) == (0, 0), "mismatches (rsq variant, v_sqrt variant)"


def test_hand_over_constant_divisions_are_the_ieee_quotients_on_this_device(built_lib):
    """The fused FSK_LDPC hand-over divides the frame's sums by Nsym = 50 and by M - 1 = 3 as x * RN(1/c) + one residual correction
    (3 instructions instead of the quotient's 11): equal to x / c for x = 0 and every float in [2^-125, FLT_MAX], counted on the device."""
    import pirip_amd
    m = pirip_amd.selftest_div()
    assert (m & 0xffffffff, m >> 32) == (0, 0), "mismatches (x / 3, x / 50)"


def test_golden_fixture_cfg1(oracle, built_lib, kernel_choice):
    g = np.load(os.path.join(GOLD, "cfg1_clean.npz"))
    _, h = _pair(oracle, sigutil.CFG1, 0, 0)
    rh = h.demod_host(g["iq_u8"])
    assert np.array_equal(rh["bits"], g["bits"])
    assert np.array_equal(rh["stats"][:, :4], g["stats"][:, :4])
    assert sigutil.rel_err(rh["rx_filt"], g["rx_filt"]) < RX_FILT_TOL


def test_golden_fixture_noisy_and_4fsk(oracle, built_lib):
    g = np.load(os.path.join(GOLD, "cfg1_noisy8dB.npz"))
    _, h = _pair(oracle, sigutil.CFG1, 0, 0)
    rh = h.demod_host(g["iq_u8"])
    assert np.array_equal(rh["bits"], g["bits"])
    assert sigutil.rel_err(rh["rx_filt"], g["rx_filt"]) < RX_FILT_TOL
    g = np.load(os.path.join(GOLD, "cfg4_clean.npz"))
    _, h = _pair(oracle, sigutil.CFG4, 0, 0)
    rh = h.demod_host(g["iq_u8"])
    assert np.array_equal(rh["bits"], g["bits"])
    assert sigutil.rel_err(rh["rx_filt"], g["rx_filt"]) < RX_FILT_TOL


def test_cfg1_600k_bit_vector_bit_exact(oracle, built_lib, kernel_choice):
    """North-star vector: 600 000 test bits, 2-FSK Fs=240k Rs=10k -p 24, u8 IQ (fsk_demod -d)."""
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 600000)
    o, h = _pair(oracle, c, oracle.IN_CU8_FSKDEMOD, 0)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
    rh = h.demod_host(u8)
    assert ro["nframes"] == 12000
    _compare(ro, rh)
    res = oracle.put_test_bits(rh["bits"], packet_pass=5990)
    assert res["errors"] == 0 and res["pass"], res
    # the smoothed spectrum after 12000 frames x 8 FFTs is bit-identical (exact FFT path)
    import ctypes as C
    Sf_o = np.ctypeslib.as_array(C.cast(_oracle_field_Sf(oracle, o), C.POINTER(C.c_float)), shape=(256,)).copy()
    assert np.array_equal(h.get_Sf(0), Sf_o)


@pytest.mark.parametrize("shape", ["wave_u8", "wave_f32_tiny", "block_u8"])
def test_estimator_root_tiers_silence_signal_and_denormal_magnitudes(oracle, built_lib, shape):
    """The estimator's |X| takes one of three forms per FFT batch (fsk_demod_wave.hip / fsk_demod_block.hip): the unguarded rsq root when no
    |X|^2 of the batch is below 2^-96 (a live receiver), the guarded one when zeros are among them (digital silence: `fsk_demod -d` maps byte
    127 to exactly 0), sqrtf for anything else (magnitudes whose squares are denormal). A stream that is silent, then carries a signal, then is
    silent again -- frames of both kinds and frames that straddle the edges -- and a complex-float stream scaled to 1e-17 (|X|^2 ~ 1e-30: normal, below 2^-96) walk all three:
    Sf bit for bit, tone estimates, frame and sample counts exact, as everywhere else."""
    import ctypes as C
    import pirip_amd
    if shape == "block_u8":
        c = dict(Fs=240000, Rs=1000, M=2, P=15, f1=11000, shift=2000, est_min=500, est_max=119000)
        fmt_o, fmt_h, ndft, quiet, nbits = oracle.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_FSKDEMOD, 4096, 30000, 150
    else:
        c = sigutil.CFG1
        fmt_o, fmt_h, ndft, quiet, nbits = oracle.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_FSKDEMOD, 256, 3000, 400
    bits = oracle.get_test_bits(nbits)
    if shape == "wave_f32_tiny":
        # (a complex-float Ts = 24 shape has no wave instance: rtl_fsk's Ts = 40 / P = 10 shape has)
        c = dict(Fs=40000, Rs=1000, M=2, P=10, f1=1000, shift=2000, est_min=500, est_max=15000)
        fmt_o, fmt_h, ndft = oracle.IN_CF32, pirip_amd.IN_CF32, 512
        x = sigutil.mod_complex(oracle, c, bits).astype(np.float32)
        z = lambda n: np.zeros((n, 2), np.float32)
        sig = np.concatenate([z(quiet), x * np.float32(1e-17), z(quiet + 700), x[:6000], z(quiet), x * np.float32(3e-20)])
    else:
        x = sigutil.mod_complex(oracle, c, bits)
        u8 = oracle.quantise_cu8(x, amp=30.0)
        sil = np.full((quiet, 2), 127, dtype=np.uint8)
        sig = np.concatenate([sil, u8, np.full((quiet + 700, 2), 127, dtype=np.uint8), u8[:len(u8) // 2], sil])
    o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], tone_spacing=100, mask=False)
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=fmt_h, nstreams=1)
    assert h.kernel() == ("block" if shape == "block_u8" else "wave"), h.kernel_name()
    ro = o.demod(sig, fmt_o)
    rh = h.demod_host(sig)
    assert ro["nframes"] >= 5
    assert rh["nframes"] == ro["nframes"] and rh["consumed"] == ro["consumed"]
    assert np.array_equal(rh["stats"][:, :4], ro["stats"][:, :4]), "tone estimates differ"
    assert np.array_equal(rh["stats"][:, 6], ro["stats"][:, 6]), "nin sequence differs"
    Sf_o = np.ctypeslib.as_array(C.cast(_oracle_field_Sf(oracle, o), C.POINTER(C.c_float)), shape=(ndft,)).copy()
    assert np.array_equal(h.get_Sf(0), Sf_o)
    if shape != "wave_f32_tiny":
        assert np.array_equal(rh["bits"], ro["bits"])


def _oracle_field_Sf(oracle, o):
    """Address of ORACLE_FSK.Sf (the oracle's own getter: no test depends on the struct's layout)."""
    import ctypes as C
    o.l.oracle_fsk_get_Sf.restype = C.c_void_p
    o.l.oracle_fsk_get_Sf.argtypes = [C.c_void_p]
    return o.l.oracle_fsk_get_Sf(o.h)


@pytest.mark.parametrize("ebno_db,seed", [(12.0, 1), (8.0, 2), (5.0, 3)])
def test_cfg1_noisy_bits_and_soft_decisions(oracle, built_lib, kernel_choice, ebno_db, seed):
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 100000, seed=seed, ebno_db=ebno_db, random_bits=True, amp=18.0)
    o, h = _pair(oracle, c, 0, 0)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
    rh = h.demod_host(u8)
    nflips = _compare(ro, rh, allow_near_tie_flips=True)
    print(f"Eb/N0 {ebno_db} dB: {nflips} near-tie bit flips of {ro['bits'].size} (margin < {2 * RX_FILT_TOL:g} of peak)")
    assert nflips <= 5


@pytest.mark.parametrize("ebno_db,seed", [(9.0, 11), (5.0, 12)])
def test_cfg4_noisy_4fsk_bits_and_soft_decisions(oracle, built_lib, kernel_choice, ebno_db, seed):
    """The near-tie contract for M = 4 (BASELINE config 4's demodulator half under noise): everything exact but the decisions the
    oracle itself takes with a margin below 2e-4 of the peak between its two largest tone magnitudes."""
    c = sigutil.CFG4
    u8, _ = sigutil.make_u8_stream(oracle, c, 100000, seed=seed, ebno_db=ebno_db, random_bits=True, amp=14.0)
    o, h = _pair(oracle, c, 0, 0)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
    rh = h.demod_host(u8)
    nflips = _compare(ro, rh, allow_near_tie_flips=True, M=4)
    print(f"4-FSK Eb/N0 {ebno_db} dB: {nflips} near-tie bit flips of {ro['bits'].size} (margin < {2 * RX_FILT_TOL:g} of peak)")
    assert nflips <= 5


def test_chunked_streaming_equals_one_shot(oracle, built_lib, kernel_choice):
    """State carries across calls: feeding ragged chunks (re-presenting the unconsumed tail)
    gives the same frames as one call and as the oracle."""
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 30000, offset=5)
    o, h = _pair(oracle, c, 0, 0)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
    rng = np.random.default_rng(0)
    pos, carry = 0, np.zeros((0, 2), dtype=np.uint8)
    bits, filt, stats = [], [], []
    while pos < u8.shape[0]:
        n = int(rng.integers(1, 5000))
        buf = np.concatenate([carry, u8[pos:pos + n]]); pos += n
        r = h.demod_host(buf)
        bits.append(r["bits"]); filt.append(r["rx_filt"]); stats.append(r["stats"])
        carry = buf[r["consumed"]:]
    rh = {"nframes": sum(len(b) for b in bits), "consumed": u8.shape[0] - carry.shape[0],
          "bits": np.concatenate(bits), "rx_filt": np.concatenate(filt), "stats": np.concatenate(stats)}
    _compare(ro, rh)


def test_edge_cases_empty_short_and_max_frames(oracle, built_lib, kernel_choice):
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 5000)
    o, h = _pair(oracle, c, 0, 0)
    r = h.demod_host(np.zeros((0, 2), dtype=np.uint8))
    assert r["nframes"] == 0 and r["consumed"] == 0
    r = h.demod_host(u8[:1199])                 # one sample short of nin: nothing happens
    assert r["nframes"] == 0 and r["consumed"] == 0
    r = h.demod_host(u8[:1200])                 # exactly one frame
    assert r["nframes"] == 1 and r["consumed"] == 1200
    ro = o.demod(u8[:1200], oracle.IN_CU8_FSKDEMOD)
    assert np.array_equal(r["bits"], ro["bits"])
    # saturated input (all 255 / all 0) and DC-only input must not crash and must match
    for fill in (255, 0, 127):
        o2, h2 = _pair(oracle, c, 0, 0)
        z = np.full((6000, 2), fill, dtype=np.uint8)
        ro = o2.demod(z, oracle.IN_CU8_FSKDEMOD); rh = h2.demod_host(z)
        assert rh["nframes"] == ro["nframes"] and np.array_equal(rh["bits"], ro["bits"])
        assert np.array_equal(rh["stats"][:, :4], ro["stats"][:, :4])


def test_sample_clock_offset_exercises_nin_feedback(oracle, built_lib, kernel_choice):
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(60000))
    n = x.shape[0]
    for ppm in (300e-6, -300e-6):
        t = np.arange(int(n / (1 + abs(ppm)) - 2)) * (1 + ppm)
        i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
        y = (1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, n - 1)]
        u8 = oracle.quantise_cu8(y)
        o, h = _pair(oracle, c, 0, 0)
        ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD); rh = h.demod_host(u8)
        assert (ro["stats"][:, 6] != 1200).any()
        _compare(ro, rh)


def test_batched_streams_device_api(oracle, built_lib, kernel_choice):
    """BASELINE config 2 shape: B independent streams (different timing offsets, tone plans and
    noise seeds) in one launch through pirip_hip_demod_batch with device pointers."""
    import torch
    import pirip_amd
    c = sigutil.CFG1
    B, nbits = 24, 5000
    streams = []
    for s in range(B):
        u8, _ = sigutil.make_u8_stream(oracle, c, nbits, seed=s, offset=s % 24, tone_bins=(s % 5) - 2,
                                       ebno_db=None if s % 3 else 10.0, random_bits=True)
        streams.append(u8)
    nsamp = min(x.shape[0] for x in streams)
    host = np.stack([x[:nsamp] for x in streams])            # [B, nsamp, 2]
    dev = torch.from_numpy(host).cuda()
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"],
                           in_format=0, nstreams=B)
    maxf = h.max_frames_for(nsamp)
    bits = torch.zeros((B, maxf, h.Nbits), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, maxf, 2 * 50), dtype=torch.float32, device="cuda")
    stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * h.Nbits, filt.data_ptr(), maxf * 100,
                  stats.data_ptr(), maxf * pirip_amd.STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), maxf, st)
    torch.cuda.synchronize()
    for s in range(B):
        o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        ro = o.demod(host[s], oracle.IN_CU8_FSKDEMOD)
        n = int(nfr[s])
        rh = {"nframes": n, "consumed": int(cons[s]), "bits": bits[s, :n].cpu().numpy(),
              "rx_filt": filt[s, :n].cpu().numpy(), "stats": stats[s, :n].cpu().numpy()}
        _compare(ro, rh)


@pytest.mark.parametrize("shape", ["ts24_u8", "ts40_s16"])
def test_wave_kernel_partial_workgroups_and_odd_strides(oracle, built_lib, shape):
    """Streams go to the wave kernel four per workgroup: stream counts that are not a multiple of four (surplus waves leave
    after the table barrier), per-stream strides that are only sample-aligned (2 bytes for u8 IQ) and a batch whose last frames differ in
    length per stream must not matter. Each stream equals its own oracle run; LDS-DMA reads past a stream's end stay inside
    the descriptor (the next stream's bytes are poisoned with a different signal)."""
    import torch
    import pirip_amd
    if shape == "ts24_u8":
        c, fmt_o, fmt_h, es = sigutil.CFG1, oracle.IN_CU8_FSKDEMOD, 0, 2
        make = lambda s: sigutil.make_u8_stream(oracle, c, 3000 + 41 * s, seed=s, offset=(5 * s) % 24, tone_bins=(s % 3) - 1, random_bits=True)[0]
    else:
        c, fmt_o, fmt_h, es = dict(sigutil.CFG3, P=8), oracle.IN_CS16, pirip_amd.IN_CS16, 4
        def make(s):
            rng = np.random.default_rng(70 + s)
            x = sigutil.mod_complex(oracle, c, rng.integers(0, 2, 700 + 13 * s).astype(np.uint8))[(7 * s) % 40:]
            return np.clip(np.trunc(x.astype(np.float64) * 5000.0), -32768, 32767).astype(np.int16)
    for B in (1, 3, 6, 7):
        streams = [make(s) for s in range(B)]
        nsamp = min(x.shape[0] for x in streams)
        stride = nsamp * es + es * (1 + B % 3)                    # bytes: a whole number of samples only (2-byte aligned for u8)
        buf = np.zeros((B, stride), dtype=np.uint8)
        for s in range(B):
            buf[s, :nsamp * es] = streams[s][:nsamp].reshape(-1).view(np.uint8)
            buf[s, nsamp * es:] = 0x55
        dev = torch.from_numpy(buf).cuda()
        h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=fmt_h, nstreams=B)
        maxf = h.max_frames_for(nsamp)
        bits = torch.zeros((B, maxf, h.Nbits), dtype=torch.uint8, device="cuda")
        stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
        cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        h.demod_batch(dev.data_ptr(), stride, nsamp, bits.data_ptr(), maxf * h.Nbits, 0, 0, stats.data_ptr(), maxf * pirip_amd.STATS_PER_FRAME,
                      nfr.data_ptr(), cons.data_ptr(), maxf, 0)
        torch.cuda.synchronize()
        for s in range(B):
            o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
            ro = o.demod(streams[s][:nsamp], fmt_o, want_filt=False)
            n = int(nfr[s])
            assert n == ro["nframes"] > 5 and int(cons[s]) == ro["consumed"]
            assert np.array_equal(bits[s, :n].cpu().numpy(), ro["bits"]), (B, s)
            assert np.array_equal(stats[s, :n, :4].cpu().numpy(), ro["stats"][:, :4]) and np.array_equal(stats[s, :n, 6].cpu().numpy(), ro["stats"][:, 6])


def test_cfg4_4fsk_and_mask_estimator(oracle, built_lib, kernel_choice):
    c = sigutil.CFG4
    u8, _ = sigutil.make_u8_stream(oracle, c, 40000, offset=2, random_bits=True, seed=4)
    o, h = _pair(oracle, c, 0, 0)
    _compare(o.demod(u8, oracle.IN_CU8_FSKDEMOD), h.demod_host(u8))
    # --mask 10000 (the reference's 4-FSK command lines: README.md:239,262)
    o, h = _pair(oracle, c, 0, 0, mask=10000)
    assert h.kernel() == ("wave" if kernel_choice == "auto" else "general")
    _compare(o.demod(u8, oracle.IN_CU8_FSKDEMOD), h.demod_host(u8))
    u8n, _ = sigutil.make_u8_stream(oracle, c, 40000, random_bits=True, seed=5, ebno_db=9.0, amp=14.0)
    o, h = _pair(oracle, c, 0, 0, mask=10000)
    _compare(o.demod(u8n, oracle.IN_CU8_FSKDEMOD), h.demod_host(u8n))


def test_cfg3_decimator_then_demod(oracle, built_lib):
    """BASELINE config 3: u8 IQ at 1.8 MS/s -> convert_u8_f | fir_decimate_cc 45 | convert_f_s16
    -> fsk_demod -c 2 40000 1000. Decimated s16 must be bit-exact, then the demod as usual."""
    import torch
    import pirip_amd
    c = sigutil.CFG3
    bits = oracle.get_test_bits(3000)
    x = sigutil.mod_complex(oracle, c, bits)                   # 40 kS/s, peak 2
    n_lo = x.shape[0]
    # x45 linear interpolation (what tlininterp does in the reference's bench Tx chain)
    t = np.arange((n_lo - 1) * 45) / 45.0
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    hi = (1 - fr) * x[i0] + fr * x[i0 + 1]
    u8 = oracle.quantise_cu8(hi, amp=40.0)
    n_in = u8.shape[0]
    # oracle chain
    L = oracle.lib()
    f = np.zeros(u8.shape, dtype=np.float32)
    L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
    ntaps = L.oracle_firdes_filter_len(0.05)
    tp = np.zeros(80, dtype=np.float32)
    L.oracle_firdes_lowpass_f_hamming(tp.ctypes.data, ntaps, 0.5 / 45)
    y = np.zeros((n_in // 45 + 1, 2), dtype=np.float32)
    n_out = L.oracle_fir_decimate_cc(f.ctypes.data, y.ctypes.data, n_in, 45, tp.ctypes.data, 80)
    s16 = np.zeros((n_out, 2), dtype=np.int16)
    L.oracle_convert_f_s16(y.ctypes.data, s16.ctypes.data, 2 * n_out)
    # HIP chain
    dec = pirip_amd.HipDecim(45, 0.05, out_s16=True)
    assert np.array_equal(dec.taps(), tp[:ntaps])
    assert dec.nout(n_in) == n_out
    d_in = torch.from_numpy(u8).cuda()
    d_out = torch.zeros((n_out, 2), dtype=torch.int16, device="cuda")
    dec.batch(d_in.data_ptr(), 0, n_in, d_out.data_ptr(), 0, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), s16), "decimated s16 differs"
    dec_f = pirip_amd.HipDecim(45, 0.05, out_s16=False)
    d_outf = torch.zeros((n_out, 2), dtype=torch.float32, device="cuda")
    dec_f.batch(d_in.data_ptr(), 0, n_in, d_outf.data_ptr(), 0, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_outf.cpu().numpy(), y[:n_out]), "decimated f32 differs"
    o, h = _pair(oracle, c, oracle.IN_CS16, 2)
    ro = o.demod(s16, oracle.IN_CS16); rh = h.demod_host(s16)
    _compare(ro, rh)
    assert oracle.put_test_bits(rh["bits"])["errors"] == 0 and rh["nframes"] >= 50
    # the two OPT-IN tap-loop arithmetics (pirip_hip_decim_set_arith): not the scalar csdr loop's rounding, so not
    # bit-exact by construction -- float outputs within a few ulp of the sum's magnitude, s16 outputs at most 1 LSB away (truncation),
    # config 3's decoded bits unchanged; back on mode 0 the stage is bit-exact again
    for mode in (1, 2):
        dec.set_arith(mode); dec_f.set_arith(mode)
        dec.batch(d_in.data_ptr(), 0, n_in, d_out.data_ptr(), 0, 1, torch.cuda.current_stream().cuda_stream)
        dec_f.batch(d_in.data_ptr(), 0, n_in, d_outf.data_ptr(), 0, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got, gotf = d_out.cpu().numpy(), d_outf.cpu().numpy()
        # (mode 2 sums byte values, ~127.5 x larger than the converted samples: its rounding steps are those of a sum near 1 + y, not near y)
        assert np.abs(gotf - y[:n_out]).max() <= (4e-7 if mode == 1 else 1.5e-6) * max(1.0, np.abs(y[:n_out]).max()), (mode, np.abs(gotf - y[:n_out]).max())
        assert np.abs(got.astype(int) - s16.astype(int)).max() <= 1, mode
        _, h2 = _pair(oracle, c, oracle.IN_CS16, 2)
        assert np.array_equal(h2.demod_host(got)["bits"], rh["bits"]), mode
    dec.set_arith(0)
    dec.batch(d_in.data_ptr(), 0, n_in, d_out.data_ptr(), 0, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), s16)


def test_cf32_and_csdr_u8_formats(oracle, built_lib):
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 8000, offset=11)
    o, h = _pair(oracle, c, oracle.IN_CU8_CSDR, 1)             # rtl_fsk's in-process convert_u8_f
    _compare(o.demod(u8, oracle.IN_CU8_CSDR), h.demod_host(u8))
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(8000))[3:]
    o, h = _pair(oracle, c, oracle.IN_CF32, 3)
    _compare(o.demod(x, oracle.IN_CF32), h.demod_host(x))


@pytest.mark.parametrize("cfgname,P", [("CFG1", 24), ("CFG1", 6), ("CFG1", 8), ("CFG4", 8)])
def test_csdr_u8_front_end_fast_instances(oracle, built_lib, kernel_choice, cfgname, P):
    """rtl_fsk's in-process convert_u8_f (x/127.5-1, /root/reference/test/loopback_rtl_fsk.sh:10, README.md:114)
    at the Ts = 24 shapes: the wave-per-stream kernel has csdr-u8 instances (P = 6 is rtl_fsk's reduced
    oversample); clean with a timing offset, AWGN, and a sample-clock offset that moves nin, on both kernels."""
    c = dict(getattr(sigutil, cfgname), P=P)
    fmt = oracle.IN_CU8_CSDR
    u8, _ = sigutil.make_u8_stream(oracle, c, 40000, offset=17, tone_bins=1)
    o, h = _pair(oracle, c, fmt, 1)
    ro = o.demod(u8, fmt); rh = h.demod_host(u8)
    assert ro["nframes"] >= 390
    _compare(ro, rh)
    u8n, _ = sigutil.make_u8_stream(oracle, c, 40000, seed=21, ebno_db=9.0, random_bits=True, amp=18.0)
    o, h = _pair(oracle, c, fmt, 1)
    if c["M"] == 2:
        assert _compare(o.demod(u8n, fmt), h.demod_host(u8n), allow_near_tie_flips=True) <= 2
    else:
        _compare(o.demod(u8n, fmt), h.demod_host(u8n))
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(40000))
    nn = x.shape[0]
    t = np.arange(int(nn / 0.9996) - 2) * 0.9996
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    y = oracle.quantise_cu8((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, nn - 1)])
    o, h = _pair(oracle, c, fmt, 1)
    ro = o.demod(y, fmt); rh = h.demod_host(y)
    assert (ro["stats"][:, 6] != 1200).any()
    _compare(ro, rh)


@pytest.mark.parametrize("shape", [
    # (Fs, Rs, M, P, format, mask spacing, f1, shift, est_max)
    (240000, 10000, 2, 6, "csdr", 10000, 10000, 10000, 60000),     # rtl_fsk ... --mask 10000 (README.md:292,297 at Ts = 24)
    (240000, 10000, 2, 8, "u8d", 10000, 10000, 10000, 60000),
    (240000, 10000, 4, 6, "csdr", 0, 10000, 10000, 60000),         # rtl_fsk -m 4: P = 6 from Ts = 24
    (240000, 10000, 4, 6, "csdr", 10000, 10000, 10000, 60000),
    (240000, 10000, 4, 8, "csdr", 10000, 10000, 10000, 60000),
    (40000, 1000, 4, 10, "cf32", 0, 1000, 2000, 18000),            # rtl_fsk -a 40000 -r 1000 -m 4 (README.md:232-239)
    (40000, 1000, 4, 10, "cf32", 2000, 1000, 2000, 18000),         # ... --mask 2000
    (40000, 1000, 4, 8, "cs16", 0, 1000, 2000, 18000),             # fsk_demod -c 4 40000 1000 behind the decimator
    (40000, 1000, 4, 8, "cs16", 2000, 1000, 2000, 18000),
    (40000, 1000, 2, 10, "cf32", 2000, 1000, 2000, 18000),
    (40000, 1000, 2, 8, "cs16", 2000, 1000, 2000, 18000),
    (200000, 10000, 4, 10, "cf32", 10000, 10000, 10000, 90000),    # rtl_fsk -a 200000 -r 10000 -m 4 --mask 10000 (README.md:262)
    (200000, 10000, 4, 10, "cf32", 0, 10000, 10000, 90000),
    (200000, 10000, 2, 10, "cf32", 10000, 10000, 10000, 90000),    # ... 2-FSK (README.md:292,297)
    (200000, 10000, 2, 10, "cf32", 0, 10000, 10000, 90000),
    (180000, 10000, 4, 9, "cf32", 10000, 10000, 10000, 80000),     # rtl_fsk -a 180000 -r 10000 -m 4 --mask 10000 (README.md:286)
    (180000, 10000, 4, 9, "cf32", 0, 10000, 10000, 80000),
    (180000, 10000, 2, 9, "cf32", 10000, 10000, 10000, 80000),
    (180000, 10000, 2, 9, "cf32", 0, 10000, 10000, 80000),
    (100000, 10000, 2, 10, "cf32", 0, 10000, 10000, 45000),        # rtl_fsk -a 100000 -r 10000 (README.md:196): Ts = 10, Ndft = 128
    (100000, 10000, 2, 10, "cf32", 10000, 10000, 10000, 45000),
    (100000, 10000, 4, 10, "cf32", 0, 10000, 10000, 48000),
    (100000, 10000, 4, 10, "cf32", 10000, 10000, 10000, 48000),
    (80000, 10000, 2, 8, "cf32", 0, 10000, 10000, 38000),          # rtl_fsk -s 2400000 -a 80000 -r 10000 (README.md:172): Ts = 8, Ndft = 128
    (80000, 10000, 2, 8, "cf32", 10000, 10000, 10000, 38000),
    (80000, 10000, 4, 8, "cf32", 0, 8000, 8000, 38000),
    (80000, 10000, 4, 8, "cf32", 8000, 8000, 8000, 38000),
], ids=lambda s: "Fs%d-M%d-P%d-%s-mask%d" % (s[0], s[2], s[3], s[4], s[5]))
def test_wave_instances_for_rtl_fsk_shapes_and_mask_estimator(oracle, built_lib, shape):
    """The instance families added for the reference's remaining command-line shapes: 4-FSK at rtl_fsk's reduced oversample
    (P = 6 / 10), 4-FSK at Ts = 40 (s16 behind the decimator, f32 inside rtl_fsk) and the `--mask` comb estimator on all of
    them (README.md:239-297). The handle must be on the wave kernel; clean, noisy (near-tie flips counted) and
    sample-clock-offset inputs against the oracle, the smoothed spectrum bit-identical."""
    import ctypes as C
    import pirip_amd
    Fs, Rs, M, P, fmtname, mask, f1, shift, est_max = shape
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1, shift=shift, est_min=500, est_max=est_max)
    Ts, Ndft = Fs // Rs, 1 << int(np.ceil(np.log2(Fs / (0.1 * Rs))))
    fmt_o, fmt_h, conv = {
        "csdr": (oracle.IN_CU8_CSDR, pirip_amd.IN_CU8_CSDR, lambda x: oracle.quantise_cu8(x, amp=18.0)),
        "u8d": (oracle.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_FSKDEMOD, lambda x: oracle.quantise_cu8(x, amp=18.0)),
        "cs16": (oracle.IN_CS16, pirip_amd.IN_CS16, lambda x: np.clip(np.trunc(x.astype(np.float64) * 8000.0), -32768, 32767).astype(np.int16)),
        "cf32": (oracle.IN_CF32, pirip_amd.IN_CF32, lambda x: np.ascontiguousarray(x * np.float32(0.37))),
    }[fmtname]
    rng = np.random.default_rng(100 + P + M + mask // 1000)
    nbits = (130 if Ts == 40 else 260) * 50 * (1 if M == 2 else 2)
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = sigutil.mod_complex(oracle, c, bits)[11:]
    o, h = _pair(oracle, c, fmt_o, fmt_h, mask=mask)
    assert h.kernel() == "wave", shape
    ro = o.demod(conv(x), fmt_o); rh = h.demod_host(conv(x))
    assert ro["nframes"] >= 120
    _compare(ro, rh)
    Sf_o = np.ctypeslib.as_array(C.cast(_oracle_field_Sf(oracle, o), C.POINTER(C.c_float)), shape=(Ndft,)).copy()
    assert np.array_equal(h.get_Sf(0), Sf_o)
    y = sigutil.add_awgn(x, 10.0 if M == 4 else 9.0, c, rng)
    o, h = _pair(oracle, c, fmt_o, fmt_h, mask=mask)
    assert _compare(o.demod(conv(y), fmt_o), h.demod_host(conv(y)), allow_near_tie_flips=True) <= 2
    nn = x.shape[0]
    t = np.arange(int(nn / 1.0004) - 2) * 1.0004
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    z = conv(((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, nn - 1)]).astype(np.float32))
    o, h = _pair(oracle, c, fmt_o, fmt_h, mask=mask)
    ro = o.demod(z, fmt_o); rh = h.demod_host(z)
    assert (ro["stats"][:, 6] != Ts * 50).any()
    _compare(ro, rh)
    # chunked: Sf, oscillator phases, integrator tail, nin and (mask) the comb position carry across calls; per-call scalars too
    o, h = _pair(oracle, c, fmt_o, fmt_h, mask=mask)
    ro = o.demod(z, fmt_o)
    got, fe, pos, carry = [], [], 0, z[:0]
    for nchunk in (3 * Ts * 50 + 7, Ts * 50 * 11, 1 << 30):
        buf = np.concatenate([carry, z[pos:pos + nchunk]]); pos += nchunk
        r = h.demod_host(buf)
        got.append(r["bits"]); fe.append(r["stats"][:, :4]); carry = buf[r["consumed"]:]
        if pos >= len(z):
            break
    assert np.array_equal(np.concatenate(got), ro["bits"]) and np.array_equal(np.concatenate(fe), ro["stats"][:, :4])


def test_reference_command_line_shapes_run_on_the_wave_kernel(built_lib):
    """Which kernel serves which configuration (pirip_hip_get_kernel): every shape the reference's command lines use at
    Fs/Rs = 24 and 40 samples per symbol is a wave-per-stream instance -- peak and mask estimator, 2- and 4-FSK, fsk_demod's
    and rtl_fsk's oversample rates, every input format those tools feed -- and anything else falls to the general kernel."""
    import pirip_amd
    A = pirip_amd
    wave = []
    for fmt in (A.IN_CU8_FSKDEMOD, A.IN_CU8_CSDR):
        wave.append((240000, 10000, 2, 24, fmt, 0))
        for M in (2, 4):
            for P in (8, 6):
                for mask in (0, 10000):
                    wave.append((240000, 10000, M, P, fmt, mask))
    for fmt in (A.IN_CS16, A.IN_CF32):
        for M in (2, 4):
            for P in (8, 10):
                for mask in (0, 2000):
                    wave.append((40000, 1000, M, P, fmt, mask))
    wave.append((48000, 1200, 2, 8, A.IN_CS16, 0))             # instances are keyed by samples per symbol, not by Fs and Rs
    for M in (2, 4):
        for mask in (0, 10000):
            wave.append((200000, 10000, M, 10, A.IN_CF32, mask))   # rtl_fsk -a 200000 -r 10000 (README.md:262,292,297)
            wave.append((180000, 10000, M, 9, A.IN_CF32, mask))    # rtl_fsk -a 180000 -r 10000 (README.md:286)
            wave.append((100000, 10000, M, 10, A.IN_CF32, mask))   # rtl_fsk -a 100000 -r 10000 (README.md:196)
            wave.append((80000, 10000, M, 8, A.IN_CF32, mask and 8000))   # rtl_fsk -s 2400000 -a 80000 -r 10000 (README.md:172)
    for Fs, Rs, M, P, fmt, mask in wave:
        h = A.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=Fs // 4, mask=mask, in_format=fmt)
        assert h.kernel() == "wave", (Fs, Rs, M, P, fmt, mask)
    general = [(160000, 10000, 4, 8, A.IN_CF32, 10000),        # 16 samples per symbol: no instance
               (200000, 10000, 4, 10, A.IN_CU8_CSDR, 10000),   # Ts = 20 with 8-bit input (the float instances serve rtl_fsk -a 200000)
               (240000, 10000, 2, 12, A.IN_CU8_FSKDEMOD, 0),   # an oversample rate nobody's command line uses
               (240000, 10000, 2, 24, A.IN_CF32, 0), (48000, 2400, 2, 10, A.IN_CS16, 0)]
    for Fs, Rs, M, P, fmt, mask in general:
        h = A.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=Fs // 4, mask=mask, in_format=fmt)
        assert h.kernel() == "general", (Fs, Rs, M, P, fmt, mask)


@pytest.mark.parametrize("fmtname,P", [("cs16", 8), ("cs16", 10), ("cf32", 8), ("cf32", 10)])
def test_ts40_ndft512_wave_instances(oracle, built_lib, kernel_choice, fmtname, P):
    """Ts = 40 / Ndft = 512 shapes: `fsk_demod -c 2 40000 1000` behind csdr's /45 decimator (README.md:109, P = 8) and the
    services' modem `rtl_fsk -a 40000 -r 1000` (script/ping:47, script/frame_repeater:36; P = 10 after the FSK_LDPC
    oversample reduction, float samples from the in-process decimator). Wave-per-stream instances vs the general kernel
    vs the oracle: clean with a timing offset, AWGN (near-tie flips counted), a sample-clock offset that moves nin, and
    chunked streaming across calls."""
    import pirip_amd
    c = dict(sigutil.CFG3, P=P)
    if fmtname == "cs16":
        fmt_o, fmt_h = oracle.IN_CS16, pirip_amd.IN_CS16
        conv = lambda x: np.clip(np.trunc(x.astype(np.float64) * 8000.0), -32768, 32767).astype(np.int16)
    else:
        fmt_o, fmt_h = oracle.IN_CF32, pirip_amd.IN_CF32
        conv = lambda x: np.ascontiguousarray(x * np.float32(0.37))
    rng = np.random.default_rng(40 + P)
    bits = rng.integers(0, 2, 6000).astype(np.uint8)
    x = sigutil.mod_complex(oracle, c, bits)[13:]
    o, h = _pair(oracle, c, fmt_o, fmt_h)
    ro = o.demod(conv(x), fmt_o); rh = h.demod_host(conv(x))
    assert ro["nframes"] >= 115
    _compare(ro, rh)
    # the smoothed spectrum after ~700 512-point FFTs is bit-identical (kiss_fft's 4,4,4,4,2 dataflow across the wave)
    import ctypes as C
    Sf_o = np.ctypeslib.as_array(C.cast(_oracle_field_Sf(oracle, o), C.POINTER(C.c_float)), shape=(512,)).copy()
    assert np.array_equal(h.get_Sf(0), Sf_o)
    y = sigutil.add_awgn(x, 9.0, c, rng)
    o, h = _pair(oracle, c, fmt_o, fmt_h)
    assert _compare(o.demod(conv(y), fmt_o), h.demod_host(conv(y)), allow_near_tie_flips=True) <= 2
    nn = x.shape[0]
    t = np.arange(int(nn / 1.0005) - 2) * 1.0005
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    z = conv(((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, nn - 1)]).astype(np.float32))
    o, h = _pair(oracle, c, fmt_o, fmt_h)
    ro = o.demod(z, fmt_o); rh = h.demod_host(z)
    assert (ro["stats"][:, 6] != 2000).any()
    _compare(ro, rh)
    # chunked: state (Sf, oscillator phase, integrator tail, nin) carries across calls
    o, h = _pair(oracle, c, fmt_o, fmt_h)
    ro = o.demod(z, fmt_o)
    got, pos, carry = [], 0, z[:0]
    for n in (7001, 4500, 12345, 1 << 30):
        buf = np.concatenate([carry, z[pos:pos + n]]); pos += n
        r = h.demod_host(buf)
        got.append(r["bits"]); carry = buf[r["consumed"]:]
        if pos >= len(z):
            break
    assert np.array_equal(np.concatenate(got), ro["bits"])


def test_cli_fsk_demod_matches_oracle_cli(oracle, built_lib):
    """Process-level boundary: the reference's command line (test/loopback_rtl_sdr.sh:16,
    README.md:105) on the product binary gives byte-identical stdout to the oracle CLI."""
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 20000, offset=9)
    raw = u8.tobytes()
    argv = ["--fsk_lower", "500", "--fsk_upper", "25000", "-d", "-p", "24", "2", "240000", "10000", "-", "-"]
    po = subprocess.run([os.path.join(ROOT, "oracle", "build", "fsk_demod_oracle")] + argv, input=raw, capture_output=True)
    ph = subprocess.run([os.path.join(BIN, "fsk_demod")] + argv, input=raw, capture_output=True)
    assert ph.returncode == 0, ph.stderr
    assert ph.stdout == po.stdout and len(ph.stdout) == 50 * (len(u8) // 1200)
    want_bits = po.stdout
    pp = subprocess.run([os.path.join(BIN, "fsk_put_test_bits"), "-q", "-p", "190", "-"], input=ph.stdout, capture_output=True)
    assert pp.returncode == 0, pp.stderr
    # soft decisions (-s) within tolerance
    po = subprocess.run([os.path.join(ROOT, "oracle", "build", "fsk_demod_oracle"), "-s"] + argv, input=raw, capture_output=True)
    ph = subprocess.run([os.path.join(BIN, "fsk_demod"), "-s"] + argv, input=raw, capture_output=True)
    a, b = np.frombuffer(ph.stdout, dtype=np.float32), np.frombuffer(po.stdout, dtype=np.float32)
    assert a.shape == b.shape and sigutil.rel_err(a, b) < RX_FILT_TOL
    # -t: one JSON object of modem statistics per frame on stderr, the bits unchanged
    import json
    pt = subprocess.run([os.path.join(BIN, "fsk_demod"), "-t"] + argv, input=raw[:2400 * 30], capture_output=True)
    assert pt.returncode == 0 and pt.stdout == want_bits[:len(pt.stdout)] and len(pt.stdout) == 50 * 30
    js = [json.loads(ln) for ln in pt.stderr.decode().split("\n") if ln.startswith("{")]
    assert len(js) == 30 and all({"EbNodB", "ppm", "f1_est", "f2_est", "eye_diagram", "samp_fft"} <= set(j) for j in js)
    assert abs(js[-1]["f1_est"] - 10312.5) < 1 and abs(js[-1]["f2_est"] - 19687.5) < 1 and js[-1]["EbNodB"] > 10
    # a slow producer (a live dongle): the bits of a frame come out as soon as the frame is in, not after a 64-frame chunk
    import time
    p = subprocess.Popen([os.path.join(BIN, "fsk_demod")] + argv, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p.stdin.write(raw[:2400 * 3]); p.stdin.flush()
    t0 = time.time(); first = p.stdout.read(100)                        # two frames' bits must arrive while stdin is still open
    assert first == want_bits[:100] and time.time() - t0 < 60
    p.stdin.write(raw[2400 * 3:2400 * 8]); p.stdin.close()
    rest = p.stdout.read(); p.wait()
    assert first + rest == want_bits[:len(first + rest)] and len(first + rest) == 50 * 8


def test_codec2_shim_single_stream(oracle, built_lib):
    """Library-level boundary (section C): fsk_create_hbr / fsk_nin / fsk_demod driven the way
    codec2's fsk_demod.c drives them."""
    import ctypes as C
    L = built_lib
    L.fsk_create_hbr.restype = C.c_void_p
    L.fsk_create_hbr.argtypes = [C.c_int] * 7
    L.fsk_set_freq_est_limits.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fsk_nin.restype = C.c_uint32; L.fsk_nin.argtypes = [C.c_void_p]
    L.fsk_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.fsk_destroy.argtypes = [C.c_void_p]
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(3000))[5:]
    fsk = L.fsk_create_hbr(c["Fs"], c["Rs"], c["M"], c["P"], 50, -1, 100)
    L.fsk_set_freq_est_limits(fsk, c["est_min"], c["est_max"])
    o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    ro = o.demod(x, oracle.IN_CF32)
    pos, out = 0, []
    while pos + L.fsk_nin(fsk) <= x.shape[0]:
        nin = L.fsk_nin(fsk)
        bits = np.zeros(50, dtype=np.uint8)
        seg = np.ascontiguousarray(x[pos:pos + nin])
        L.fsk_demod(fsk, bits.ctypes.data, seg.ctypes.data)
        out.append(bits); pos += nin
    L.fsk_destroy(fsk)
    assert np.array_equal(np.stack(out), ro["bits"])


def test_codec2_shim_clear_estimators_like_upstream(oracle, built_lib):
    """fsk_clear_estimators() in the middle of a stream zeroes the smoothed spectrum and puts nin back to N -- oscillator
    phases, integrator memory and timing estimates stay, as upstream leaves them: the bits and tone estimates of the frames
    that follow equal the oracle's, which restates exactly that."""
    import ctypes as C
    L = built_lib
    L.fsk_create_hbr.restype = C.c_void_p
    L.fsk_create_hbr.argtypes = [C.c_int] * 7
    L.fsk_set_freq_est_limits.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fsk_nin.restype = C.c_uint32; L.fsk_nin.argtypes = [C.c_void_p]
    L.fsk_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.fsk_clear_estimators.argtypes = [C.c_void_p]
    L.fsk_get_f_est.argtypes = [C.c_void_p, C.c_void_p]
    L.fsk_destroy.argtypes = [C.c_void_p]
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(4000))[9:]
    x = sigutil.add_awgn(x, 12.0, c, np.random.default_rng(8))
    fsk = L.fsk_create_hbr(c["Fs"], c["Rs"], c["M"], c["P"], 50, -1, 100)
    L.fsk_set_freq_est_limits(fsk, c["est_min"], c["est_max"])
    o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    pos, frame = 0, 0
    while pos + L.fsk_nin(fsk) <= x.shape[0]:
        if frame in (7, 30):
            L.fsk_clear_estimators(fsk); o.clear_estimators()
        nin = L.fsk_nin(fsk)
        assert nin == o.nin()
        seg = np.ascontiguousarray(x[pos:pos + nin])
        bits = np.zeros(50, dtype=np.uint8)
        L.fsk_demod(fsk, bits.ctypes.data, seg.ctypes.data)
        ro = o.demod(seg, oracle.IN_CF32)
        assert ro["nframes"] == 1 and np.array_equal(bits, ro["bits"][0]), frame
        fe = np.zeros(4, dtype=np.float32)
        L.fsk_get_f_est(fsk, fe.ctypes.data)
        assert np.array_equal(fe[:2], ro["stats"][0, :2]), frame
        pos += nin; frame += 1
    L.fsk_destroy(fsk)
    assert frame > 60


def test_rtl_fsk_dashboard_json_over_udp(oracle, built_lib):
    """`rtl_fsk ... -u host` (test/loopback_rtl_fsk.sh:10, README.md:119,123): once per second of samples one JSON object per
    datagram to port 8001 with exactly the keys script/dash.py:26-45 reads, and value ranges it plots (timing within +-0.5,
    Ndft spectrum bins, tone estimates between the limits)."""
    import json
    import socket
    sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        sock.bind(("127.0.0.1", 8001))
    except OSError:
        pytest.skip("UDP port 8001 is taken on this box")
    sock.settimeout(20)
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 30000, offset=5)            # 3 s at 240 kS/s
    p = subprocess.run([os.path.join(BIN, "rtl_fsk"), "-g", "1", "-s", "240000", "-f", "144480000", "-", "-n", str(u8.shape[0]),
                        "-u", "127.0.0.1"], input=u8.tobytes(), capture_output=True, env=dict(os.environ, PIRIP_IQ_FILE="/dev/stdin"))
    assert p.returncode == 0, p.stderr
    msgs = []
    try:
        while len(msgs) < 2:
            msgs.append(sock.recvfrom(65536)[0])
    except socket.timeout:
        pass
    sock.close()
    assert len(msgs) >= 2
    for raw in msgs:
        assert raw.endswith(b"\n")
        d = json.loads(raw)
        assert set(d) == {"SNRest_lin", "norm_rx_timing", "SfdB", "fsk_lower_Hz", "fsk_upper_Hz", "f_est_Hz", "Fs_Hz"}
        assert d["Fs_Hz"] == 240000 and len(d["SfdB"]) == 256 and len(d["f_est_Hz"]) == 2
        assert 150 <= len(d["norm_rx_timing"]) <= 250 and all(-0.5 <= t <= 0.5 for t in d["norm_rx_timing"])   # ~200 frames per second
        assert d["fsk_lower_Hz"] < d["f_est_Hz"][0] < d["f_est_Hz"][1] < d["fsk_upper_Hz"] and d["SNRest_lin"] > 10
    assert oracle.put_test_bits(np.frombuffer(p.stdout, dtype=np.uint8))["errors"] == 0


def test_cpp_multi_gpu_receiver_single_rank(built_lib):
    """The C++ host of the multi-GPU path (C-ABI + pirip_hip_gather_bits over RCCL, include/pirip_hip_rccl.h) at world
    size 1: tools/launch_mgpu.sh is what runs it at 2/4/8. Every gathered bit is a transmitted test bit."""
    import json
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_mgpu.sh"), "1", "--streams", "96", "--samples", "240000", "--steps", "2", "--warmup", "1"],
                       capture_output=True, env=dict(os.environ, PIRIP_MGPU="cpp"), timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.decode().split("\n") if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and 96 * 199 <= d["frames_gathered_per_step"] <= 96 * 200 and d["bit_errors_vs_tx"] == 0 and d["test_bits_checked"] >= 8000


def test_library_boundary_c_program_written_like_upstream(oracle, built_lib, tmp_path):
    """SURVEY.md 8b library level: a plain-C receive loop written the way codec2's fsk_demod.c / rtl_fsk.c use libcodec2 --
    direct reads of struct FSK fields, full-layout MODEM_STATS, libcsdr's firdes_lowpass_f(.., window_t) -- compiled
    against include/pirip_hip.h only and linked to libpirip_hip.so (what /root/reference/build_rtlsdr.sh:9 links).
    Bits, tone estimates and nin are the oracle's; timing / SNR figures within tolerance; snr_est is the smoothed EbNodB."""
    import pirip_amd
    exe = str(tmp_path / "fsk_like")
    libdir = os.path.dirname(pirip_amd.lib_path())
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cprog", "fsk_demod_like_upstream.c"), "-L", libdir, "-lpirip_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    c = dict(Fs=48000, Rs=1200, M=2, P=8, f1=1200, shift=1200, est_min=300, est_max=6000)     # Ts = 40, CF32 input
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, 4000).astype(np.uint8)
    x = sigutil.add_awgn(sigutil.mod_complex(oracle, c, bits)[7:], 12.0, c, rng)
    s16 = np.clip(np.trunc(x.astype(np.float64) * 6000.0), -32768, 32767).astype(np.int16)
    p = subprocess.run([exe, "2", "48000", "1200", "8", "300", "6000", "eye"], input=s16.tobytes(), capture_output=True)
    assert p.returncode == 0, p.stderr[-2000:]
    # the same program without the eye opt-in stays on the specialised wave instance: same bits, no traces,
    # and the environment switch turns them on without a source change
    env = {k: v for k, v in os.environ.items() if k != "PIRIP_SHIM_EYE"}
    p_fast = subprocess.run([exe, "2", "48000", "1200", "8", "300", "6000"], input=s16.tobytes(), capture_output=True, env=env)
    assert p_fast.returncode == 0, p_fast.stderr[-2000:]
    assert p_fast.stdout == p.stdout
    assert all(" neyetr 0 " in ln for ln in p_fast.stderr.decode().split("\n") if " nin " in ln)
    p_env = subprocess.run([exe, "2", "48000", "1200", "8", "300", "6000"], input=s16.tobytes(), capture_output=True, env=dict(env, PIRIP_SHIM_EYE="1"))
    assert p_env.returncode == 0 and p_env.stdout == p.stdout and p_env.stderr == p.stderr
    o = oracle.OracleFsk(c["Fs"], c["Rs"], 2, P=8, est_min=300, est_max=6000)
    # frame by frame, as the program does, so that the oracle's per-frame by-products line up
    pos, want_bits, rows = 0, [], []
    while pos + o.nin() <= len(s16):
        n = o.nin()
        r = o.demod(s16[pos:pos + n], oracle.IN_CS16)
        assert r["nframes"] == 1
        want_bits.append(r["bits"][0]); rows.append(np.concatenate([r["stats"][0], o.snr(), [o.eye().astype(np.float64).sum()]])); pos += n
    want_bits = np.concatenate(want_bits); rows = np.array(rows)
    assert p.stdout == want_bits.tobytes()
    lines = [ln.split() for ln in p.stderr.decode().split("\n") if " nin " in ln]
    assert len(lines) == len(rows) > 30
    for ln, w in zip(lines, rows):
        f = {k: ln[ln.index(k) + 1] for k in ("nin", "timing", "SNRest", "ppm", "EbNodB", "snr_est", "clock", "rx_timing", "sfpeak", "neyetr", "neyesamp", "eyesum", "eyemax")}
        assert int(f["nin"]) == int(w[6])
        assert float(ln[ln.index("f_est") + 1]) == pytest.approx(float(w[0]), abs=1e-3) and float(ln[ln.index("f_est") + 2]) == pytest.approx(float(w[1]), abs=1e-3)
        assert abs(float(f["timing"]) - float(w[4])) < TIMING_TOL
        assert float(f["SNRest"]) == pytest.approx(float(w[5]), rel=SNR_TOL)
        ns = pirip_amd.STATS_PER_FRAME                       # (the oracle's snr_est, EbNodB, v_est follow the stats row)
        assert float(f["EbNodB"]) == pytest.approx(float(w[ns + 1]), abs=2e-2) and float(f["snr_est"]) == pytest.approx(float(w[ns]), abs=2e-2)
        assert float(f["rx_timing"]) == pytest.approx(float(w[4]) * 8, abs=1e-3)
        # MODEM_STATS eye diagram of the frame: 8 / M traces per tone, two symbols (2P = 16 points) each, normalised to 1
        assert (int(f["neyetr"]), int(f["neyesamp"])) == (8, 16) and float(f["eyemax"]) == pytest.approx(1.0, abs=1e-6)
        assert float(f["eyesum"]) == pytest.approx(float(w[-1]), rel=2e-4)
        assert float(f["clock"]) == pytest.approx(float(w[7]), abs=0.5)
        assert abs(int(f["sfpeak"]) - 256 - 1200 * 512 // 48000) <= 14       # Sf host copy is live: its peak sits on one of the tones


@pytest.mark.parametrize("cfg,fmt,nbits", [
    (dict(sigutil.CFG1), "u8", 6000),                                                                   # P = 24: 48 points per trace
    (dict(sigutil.CFG4), "u8", 6000),                                                                   # 4-FSK: 2 traces per tone
    (dict(Fs=96000, Rs=1000, M=2, P=96, f1=4000, shift=2000, est_min=500, est_max=12000), "u8", 800),    # 2P = 192 > 160: every 2nd position
    (dict(Fs=48000, Rs=1200, M=2, P=8, f1=1200, shift=1200, est_min=300, est_max=6000), "u8", 3000),
])
def test_eye_diagram_of_the_latest_frame_matches_oracle(oracle, built_lib, cfg, fmt, nbits):
    """MODEM_STATS.rx_eye (what `fsk_demod -t` plots): after a call the handle holds |f_int| eye traces of the last frame of each
    stream -- rows, points per row and values are the oracle's (values to the correlator tolerance), normalised and raw."""
    import pirip_amd
    c = cfg
    u8, _ = sigutil.make_u8_stream(oracle, c, nbits, seed=4, ebno_db=12.0, random_bits=True, amp=20.0, offset=5)
    o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=0)
    with pytest.raises(RuntimeError):
        h.eye()                                             # not enabled: refused, not zeros
    h.enable_eye()
    assert h.kernel() == "general"
    # two calls (the traces follow the latest frame), the unconsumed tail carried like a reader would
    carry = np.zeros((0, 2), dtype=np.uint8)
    for piece in (u8[: len(u8) // 2], u8[len(u8) // 2:]):
        buf = np.concatenate([carry, piece])
        ro = o.demod(buf, oracle.IN_CU8_FSKDEMOD)
        rh = h.demod_host(buf)
        assert rh["nframes"] == ro["nframes"] > 0 and rh["consumed"] == ro["consumed"]
        assert np.array_equal(rh["bits"], ro["bits"])
        carry = buf[ro["consumed"]:]
        eo_raw, eh_raw = o.eye(normalise=False), h.eye(normalise=False)
        M, P = c["M"], c["P"]
        dec = -(-2 * P // 160)
        assert eh_raw.shape == eo_raw.shape == ((8 // M) * M, 2 * P // dec)
        assert np.abs(eh_raw - eo_raw).max() <= RX_FILT_TOL * eo_raw.max()
        eo, eh = o.eye(), h.eye()
        assert eh.max() == 1.0 and np.abs(eh - eo).max() <= 2 * RX_FILT_TOL
    h.close()


def test_fsk_demod_testmode_json_carries_eye_diagram_and_spectrum(oracle, built_lib):
    """`fsk_demod -t`: one JSON object per frame on stderr -- [UPSTREAM-RECALLED fsk_demod.c] "eye_diagram" is MODEM_STATS.rx_eye
    (neyetr rows of neyesamp points, normalised) and "samp_fft" the first Ndft/2 values of the estimator spectrum Sf."""
    import json
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 1000, seed=2, ebno_db=12.0, random_bits=True, amp=20.0)
    p = subprocess.run([os.path.join(BIN, "fsk_demod"), "-d", "-p", "24", "-t", "2", "240000", "10000", "-", "-"], input=u8.tobytes(), capture_output=True)
    assert p.returncode == 0, p.stderr[-2000:]
    o = oracle.OracleFsk(c["Fs"], c["Rs"], 2, P=24)
    objs = [json.loads(ln) for ln in p.stderr.decode().split("\n") if ln.startswith("{")]
    pos, k, bits = 0, 0, []
    while pos + o.nin() <= len(u8):
        n = o.nin()
        r = o.demod(u8[pos:pos + n], oracle.IN_CU8_FSKDEMOD); pos += n
        bits.append(r["bits"][0])
        j = objs[k]; k += 1
        eye = np.array(j["eye_diagram"], dtype=np.float32)
        assert eye.shape == (8, 48) and np.abs(eye - o.eye()).max() < 5e-4        # %f: six decimals
        assert len(j["samp_fft"]) == 128
        assert j["f1_est"] == pytest.approx(float(r["stats"][0, 0]), abs=0.06) and j["f2_est"] == pytest.approx(float(r["stats"][0, 1]), abs=0.06)
    assert k == len(objs) > 15 and p.stdout == np.concatenate(bits).tobytes()


def test_rtl_fsk_cli_direct_and_decimated(oracle, built_lib):
    """rtl_fsk packaging: the reference's integrated receiver command line
    (test/loopback_rtl_fsk.sh:10) with a file in place of the dongle; in-process convert_u8_f
    (x/127.5-1) -> fsk_demod, and with -a a decimate-then-demod chain (README.md:172)."""
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 30000, offset=3)
    exe = os.path.join(BIN, "rtl_fsk")
    # (1) direct: defaults Fs 240k Rs 10k M 2; P follows the halving rule (24 -> 12 -> 6)
    p = subprocess.run([exe, "-g", "1", "-s", "240000", "-f", "144480000", "-i", "-", "-", "-n", str(u8.shape[0]),
                        "-l", "500", "-U", "25000"], input=u8.tobytes(), capture_output=True)
    assert p.returncode == 0, p.stderr
    o = oracle.OracleFsk(240000, 10000, 2, P=6, est_min=500, est_max=25000)
    ro = o.demod(u8, oracle.IN_CU8_CSDR, want_filt=False)
    assert p.stdout == ro["bits"].tobytes()
    pp = subprocess.run([os.path.join(BIN, "fsk_put_test_bits"), "-q", "-p", "290", "-"], input=p.stdout, capture_output=True)
    assert pp.returncode == 0, pp.stderr
    # (2) decimated: 1.2 MS/s u8 -> /5 -> 240 kS/s modem rate
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(6000))
    n_lo = x.shape[0]
    t = np.arange((n_lo - 1) * 5) / 5.0
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    hi = oracle.quantise_cu8((1 - fr) * x[i0] + fr * x[i0 + 1], amp=40.0)
    p = subprocess.run([exe, "-s", "1200000", "-a", "240000", "-r", "10000", "-i", "-", "-", "-l", "500", "-U", "25000"],
                       input=hi.tobytes(), capture_output=True)
    assert p.returncode == 0, p.stderr
    got = np.frombuffer(p.stdout, dtype=np.uint8)
    res = oracle.put_test_bits(got)
    assert res["errors"] == 0 and res["packets"] >= 50, res


@pytest.mark.parametrize("P", [8, 6])
def test_other_oversample_rates_fast_instances(oracle, built_lib, kernel_choice, P):
    """`fsk_demod -d 2 240000 10000` (default P = 8) and rtl_fsk's P = 6: the specialised kernel has
    instances for them; clean + sample-clock-offset + noisy streams, both kernels."""
    c = dict(sigutil.CFG1, P=P)
    u8, _ = sigutil.make_u8_stream(oracle, c, 40000, offset=13)
    o, h = _pair(oracle, c, 0, 0)
    _compare(o.demod(u8, oracle.IN_CU8_FSKDEMOD), h.demod_host(u8))
    u8n, _ = sigutil.make_u8_stream(oracle, c, 40000, seed=9, ebno_db=9.0, random_bits=True, amp=18.0)
    o, h = _pair(oracle, c, 0, 0)
    n = _compare(o.demod(u8n, oracle.IN_CU8_FSKDEMOD), h.demod_host(u8n), allow_near_tie_flips=True)
    assert n <= 2
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(40000))
    nn = x.shape[0]
    t = np.arange(int(nn / 1.0004) - 2) * 1.0004
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    y = oracle.quantise_cu8((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, nn - 1)])
    o, h = _pair(oracle, c, 0, 0)
    ro = o.demod(y, oracle.IN_CU8_FSKDEMOD); rh = h.demod_host(y)
    assert (ro["stats"][:, 6] != 1200).any()
    _compare(ro, rh)


@pytest.mark.parametrize("Fs,Rs,M,P,f1,shift", [
    (48000, 1200, 2, 8, 1200, 1200),      # Ts=40  Ndft=512  (4,4,4,4,2 factorisation)
    (96000, 2400, 4, 10, 2400, 2400),     # Ts=40  P=10, 4-FSK
    (8000, 100, 2, 8, 300, 200),          # Ts=80  Ndft=1024, nold up to 180 (> one wave)
    (240000, 10000, 2, 12, 10000, 10000), # Ts=24  P=12: no specialised instance -> general kernel
    (80000, 10000, 2, 8, 10000, 10000),   # Ts=8   Ndft=128: rtl_fsk's "-a 80000" modem rate (README.md:172)
    (200000, 10000, 4, 5, 10000, 20000),  # Ts=20  P=5: README.md:262's 200 kHz / 4-FSK plan
    (40000, 1000, 2, 10, 1000, 1000),     # Ts=40  P=10: the services' modem, rtl_fsk -a 40000 -r 1000 (script/ping:6,47)
    (240000, 1000, 2, 15, 11000, 2000),   # Ts=240 P=15 Ndft=4096: rtl_fsk -r 1000 at 240 kS/s (README.md:152,184); groups of 4 samples
    (240000, 1000, 4, 15, 11000, 2000),   #        ... -m 4 (README.md:239)
])
def test_general_kernel_configuration_sweep(oracle, built_lib, kernel_choice, Fs, Rs, M, P, f1, shift):
    """The configuration space fsk_create_hbr accepts against the oracle: other FFT sizes (radix-2 leaf), other Ts/P, 4-FSK, long
    integrator memories; clean + AWGN. Once on the kernel the library picks (a wave instance, the Ts = 240 block instance, or the
    general kernel) and once with the general kernel forced."""
    import pirip_amd
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1, shift=shift, est_min=Rs // 2, est_max=Fs // 2 - Rs)
    nbits = 6000 * (1 if M == 2 else 2)
    rng = np.random.default_rng(Fs + M)
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = sigutil.mod_complex(oracle, c, bits)[rng.integers(0, Fs // Rs):]
    for ebno in (None, 10.0):
        y = x if ebno is None else sigutil.add_awgn(x, ebno, c, rng)
        u8 = oracle.quantise_cu8(y, amp=20.0)
        o = oracle.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"])
        h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], in_format=0, nstreams=1)
        ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD); rh = h.demod_host(u8)
        assert ro["nframes"] > 50
        # Soft-decision tolerance: 1e-4 of the peak for frames of up to ~2400 samples. Upstream's oscillators are float32
        # recursions whose magnitude wanders by ~6e-8 per sample; the kernels follow that drift to first order, but a tone at an
        # exact short-period fraction of Fs (15 kHz at 240 kS/s: period 16) locks the rounded recursion into a limit cycle after
        # a few thousand samples instead of drifting on -- start-phase dependent, not modelled. Over 12 000-sample frames
        # (Ts = 240) that is 2e-4 of the peak, on whole symbols alike (decisions are unaffected): the bar scales with N.
        tol = RX_FILT_TOL * max(1.0, (Fs // Rs) * 50 / 2400.0)
        if M == 2:
            _compare(ro, rh, tol=tol, allow_near_tie_flips=ebno is not None)
        else:
            _compare(ro, rh, tol=tol)


def test_max_frames_limit_and_resume_on_device(oracle, built_lib, kernel_choice):
    """pirip_hip_demod_batch stops after max_frames; re-presenting the unconsumed tail (device
    pointer advanced by d_consumed) resumes exactly where it stopped -- B streams, 3 frames per call."""
    import torch
    import pirip_amd
    c = sigutil.CFG1
    B = 5
    host = []
    for s in range(B):
        u8, _ = sigutil.make_u8_stream(oracle, c, 1200, seed=s, offset=3 * s, random_bits=True)
        host.append(u8[:28000])
    host = np.stack(host)
    nsamp = host.shape[1]
    dev = torch.from_numpy(host).cuda()
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], nstreams=B)
    bits = torch.zeros((B, 3, 50), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    got = [[] for _ in range(B)]
    off = np.zeros(B, dtype=np.int64)          # per-stream resume point (nin sequences differ per stream)
    for _ in range(40):
        L = int((nsamp - off).min())
        if L < 1206:
            break
        # re-present every stream's unconsumed tail at the front of a fresh batch buffer
        batch = torch.stack([dev[s, int(off[s]):int(off[s]) + L] for s in range(B)]).contiguous()
        h.demod_batch(batch.data_ptr(), L * 2, L, bits.data_ptr(), 150, 0, 0, 0, 0,
                      nfr.data_ptr(), cons.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(nfr.max()) <= 3
        for s in range(B):
            got[s].append(bits[s, :int(nfr[s])].cpu().numpy().copy())
        off += cons.cpu().numpy()
    for s in range(B):
        o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        ro = o.demod(host[s], oracle.IN_CU8_FSKDEMOD, want_filt=False)
        g = np.concatenate(got[s])
        assert g.shape[0] >= 20 and np.array_equal(g, ro["bits"][:g.shape[0]])


def test_burst_mode_pins_nin(oracle, built_lib, kernel_choice):
    """fsk_enable_burst_mode(): nin stays N although the timing estimate crosses +-0.25."""
    c = sigutil.CFG1
    x = sigutil.mod_complex(oracle, c, oracle.get_test_bits(30000))
    n = x.shape[0]
    t = np.arange(int(n / 1.0005) - 2) * 1.0005
    i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
    u8 = oracle.quantise_cu8((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, n - 1)])
    o, h = _pair(oracle, c, 0, 0)
    o.enable_burst_mode(); h.set_burst_mode(True)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD); rh = h.demod_host(u8)
    assert (ro["stats"][:, 6] == 1200).all() and (np.abs(ro["stats"][:, 4]) > 0.25).any()
    _compare(ro, rh)


def test_cli_csdr_three_process_pipe(oracle, built_lib):
    """Process-level boundary of the wide-band form (README.md:109):
        csdr convert_u8_f | csdr fir_decimate_cc 45 | csdr convert_f_s16
    each stage as its own process on the product binaries; the s16 stream must equal the oracle's
    restatement of csdr's block loop (16384-sample blocks, trailing partial block dropped)."""
    rng = np.random.default_rng(21)
    n = 16384 + 16335 * 3 + 5000
    u8 = rng.integers(0, 256, (n, 2)).astype(np.uint8)
    exe = os.path.join(BIN, "csdr")
    p1 = subprocess.run([exe, "convert_u8_f"], input=u8.tobytes(), capture_output=True)
    assert p1.returncode == 0, p1.stderr
    f = np.frombuffer(p1.stdout, dtype=np.float32)
    L = oracle.lib()
    fo = np.zeros(u8.size, dtype=np.float32)
    L.oracle_convert_u8_f(u8.ctypes.data, fo.ctypes.data, u8.size)
    assert np.array_equal(f, fo)
    p2 = subprocess.run([exe, "fir_decimate_cc", "45"], input=p1.stdout, capture_output=True)
    assert p2.returncode == 0, p2.stderr
    y = np.frombuffer(p2.stdout, dtype=np.float32).reshape(-1, 2)
    yo = np.zeros((4096, 2), dtype=np.float32)
    no = L.oracle_csdr_fir_decimate_stream(fo.ctypes.data, n, yo.ctypes.data, 4096, 45, 0.05, 16384)
    assert no == 4 * 363 and y.shape[0] == no
    assert np.array_equal(y, yo[:no])
    p3 = subprocess.run([exe, "convert_f_s16"], input=p2.stdout, capture_output=True)
    s16 = np.frombuffer(p3.stdout, dtype=np.int16)
    so = np.zeros(2 * no, dtype=np.int16)
    L.oracle_convert_f_s16(yo.ctypes.data, so.ctypes.data, 2 * no)
    assert np.array_equal(s16, so)


def test_cli_real_and_complex_s16_inputs(oracle, built_lib):
    """fsk_demod's other stdin formats: real s16 (no -c/-d; tone search limited to 0..Fs/2) and complex
    s16 (-c, README.md:109), fed by the product's own fsk_mod exactly as the reference's bench chain does
    (fsk_get_test_bits | fsk_mod [-c] 2 40000 1000 1000 2000, README.md:142)."""
    bits = subprocess.run([os.path.join(BIN, "fsk_get_test_bits"), "-", "6000"], capture_output=True).stdout
    for flag in ([], ["-c"]):
        mod = subprocess.run([os.path.join(BIN, "fsk_mod")] + flag + ["2", "40000", "1000", "1000", "2000", "-", "-"],
                             input=bits, capture_output=True)
        assert mod.returncode == 0 and len(mod.stdout) == 6000 * 40 * 2 * (2 if flag else 1)
        argv = flag + ["--fsk_lower", "500", "2", "40000", "1000", "-", "-"]
        po = subprocess.run([os.path.join(ROOT, "oracle", "build", "fsk_demod_oracle")] + argv, input=mod.stdout, capture_output=True)
        ph = subprocess.run([os.path.join(BIN, "fsk_demod")] + argv, input=mod.stdout, capture_output=True)
        assert ph.returncode == 0, ph.stderr
        assert ph.stdout == po.stdout and len(ph.stdout) >= 50 * 100
        pp = subprocess.run([os.path.join(BIN, "fsk_put_test_bits"), "-q", "-p", "55", "-"], input=ph.stdout, capture_output=True)
        assert pp.returncode == 0, pp.stderr


@pytest.mark.parametrize("cfgname", ["CFG1", "CFG4"])
def test_packed_bit_output_equals_packbits(oracle, built_lib, kernel_choice, cfgname):
    """pirip_hip_set_bit_packing(): 8 bits per byte, MSB first == numpy.packbits of the one-byte-per-bit
    output (2-FSK: 50 bits -> 7 bytes per frame, 4-FSK: 100 bits -> 13 bytes)."""
    c = getattr(sigutil, cfgname)
    u8, _ = sigutil.make_u8_stream(oracle, c, 20000, offset=5, random_bits=True, seed=12)
    _, h = _pair(oracle, c, 0, 0)
    ref = h.demod_host(u8)["bits"]
    _, hp = _pair(oracle, c, 0, 0)
    hp.set_bit_packing(True)
    got = hp.demod_host(u8)["bits"]
    assert got.shape == (ref.shape[0], (ref.shape[1] + 7) // 8)
    assert np.array_equal(got, np.packbits(ref, axis=1))


def test_api_argument_validation_and_many_small_streams(oracle, built_lib):
    import ctypes as C
    import torch
    import pirip_amd
    c = sigutil.CFG1
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], nstreams=3000)
    L = built_lib
    assert L.pirip_hip_demod_batch(h.h, None, 0, 100, None, 0, None, 0, None, 0, None, None, 1, None) == -1   # NULL input
    assert L.pirip_hip_demod_batch(None, 1, 0, 100, None, 0, None, 0, None, 0, None, None, 1, None) == -1     # NULL handle
    assert L.pirip_hip_demod_batch(h.h, 1, 0, -5, None, 0, None, 0, None, 0, None, None, 1, None) == -1       # negative size
    assert L.pirip_hip_get_Sf(h.h, 3000, 1) == -1 and L.pirip_hip_get_info(h.h, None) == -1
    assert L.pirip_hip_strerror(-3) == b"no usable HIP device"
    # 3000 streams x 3 frames each, every stream a different slice of one long signal, outputs NULL except bits
    u8, _ = sigutil.make_u8_stream(oracle, c, 3000 + 200, random_bits=True, seed=31)
    nsamp = 3700
    idx = (np.arange(3000) * 7)[:, None] + np.arange(nsamp)[None, :]
    host = u8[idx]                                            # [3000, nsamp, 2]
    dev = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    bits = torch.zeros((3000, 4, 50), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(3000, dtype=torch.int32, device="cuda")
    h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), 200, d_nframes=nfr.data_ptr(), max_frames=4,
                  stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(nfr.min()) == 3 and int(nfr.max()) == 3
    for s in (0, 1, 1499, 2999):
        o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        ro = o.demod(host[s], oracle.IN_CU8_FSKDEMOD, want_filt=False)
        assert np.array_equal(bits[s, :3].cpu().numpy(), ro["bits"][:3])


def _synth(c, f1s, skips, bits, nsamp, amp=32.0, sigma=0.0, seed=1):
    import torch
    import pirip_amd
    from pirip_amd.binding import synth_cu8
    B = len(f1s)
    bps = 1 if c["M"] == 2 else 2
    dbits = torch.from_numpy(np.ascontiguousarray(bits)).cuda()
    out = torch.zeros((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    stride = 0 if bits.ndim == 1 else bits.shape[1]
    nsym = bits.shape[-1] // bps
    synth_cu8(c["Fs"], c["Rs"], c["M"], f1s, c["shift"], dbits.data_ptr(), stride, nsym, out.data_ptr(), nsamp * 2,
              nsamp, amp=amp, sigma=sigma, seed=seed, skip=skips, stream=torch.cuda.current_stream().cuda_stream)
    return out


@pytest.mark.parametrize("cfgname,amp", [("CFG1", 32.0), ("CFG4", 32.0), ("CFG1", 40.5)])
def test_device_synth_matches_cpu_modulator(oracle, built_lib, cfgname, amp):
    """the device-side Tx (fsk_mod -c | u8 quantiser) is bit-identical to the
    oracle's modulator called the way the fsk_mod tool calls it (50 symbols per call)."""
    c = getattr(sigutil, cfgname)
    bps = 1 if c["M"] == 2 else 2
    B, nsym = 5, 2030
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, (B, nsym * bps)).astype(np.uint8)
    f1s = [c["f1"] + d for d in (0, 937, -1875, 333, 2500)]
    skips = [0, 1, 13, 23, 100]
    ts = c["Fs"] // c["Rs"]
    nsamp = nsym * ts - 100
    got = _synth(c, f1s, skips, bits, nsamp, amp=amp).cpu().numpy()
    for s in range(B):
        tx = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], f1_tx=f1s[s], tone_spacing=c["shift"])
        x = np.concatenate([tx.mod_c(bits[s, i:i + 50 * bps]) for i in range(0, nsym * bps, 50 * bps)])
        q = np.rint(np.float32(127.0) + np.float32(amp) * x.astype(np.float32))       # float32 arithmetic
        assert q.dtype == np.float32
        want = np.clip(q, 0, 255).astype(np.uint8)[skips[s]:skips[s] + nsamp]
        assert np.array_equal(got[s], want), (s, int((got[s] != want).sum()))
    # a shared bit vector (bits_stride 0) gives every stream the same symbols
    got0 = _synth(c, [c["f1"]] * 3, None, bits[0], nsamp, amp=amp).cpu().numpy()
    assert np.array_equal(got0[0], got0[2]) and np.array_equal(got0[0], _synth(c, [c["f1"]], None, bits[:1], nsamp, amp=amp).cpu().numpy()[0])


def test_device_synth_awgn_statistics_and_ber(oracle, built_lib):
    """The on-device AWGN source: per-component variance and kurtosis of the residual, independence
    across streams, and an Eb/N0 = 8 dB BER through the HIP demodulator against the non-coherent
    2-FSK formula 0.5*exp(-Eb/2N0) (implementation loss of the 8-bit front end and the timing/tone
    estimators allowed for: [0.8, 1.6] x theory)."""
    import torch
    import pirip_amd
    c = sigutil.CFG1
    B, nbits = 64, 20000
    ts = c["Fs"] // c["Rs"]
    nsamp = nbits * ts
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2, (B, nbits)).astype(np.uint8)
    ebno = 10 ** 0.8
    sigma = float(np.sqrt((4.0 * ts / ebno) / 2.0))
    amp = 8.0          # 5.8 sigma of headroom either side of mid-scale
    f1s = [c["f1"]] * B
    clean = _synth(c, f1s, None, bits, nsamp, amp=amp)
    noisy = _synth(c, f1s, None, bits, nsamp, amp=amp, sigma=sigma, seed=99)
    assert int(((noisy == 0) | (noisy == 255)).sum()) < 1e-4 * noisy.numel()       # no meaningful clipping
    r = (noisy[:8].cpu().numpy().astype(np.float64) - clean[:8].cpu().numpy().astype(np.float64)) / amp
    var = r.reshape(-1, 2).var(axis=0)
    want_var = sigma ** 2 + (1.0 / 12.0 + 1.0 / 12.0) / amp ** 2                      # two roundings
    assert np.all(np.abs(var / want_var - 1) < 0.01), (var, want_var)
    assert abs(r.mean()) < 5e-3 * sigma
    kurt = np.mean(r ** 4) / np.mean(r ** 2) ** 2
    assert abs(kurt - 3.0) < 0.02, kurt
    assert abs(np.corrcoef(r[0, :, 0], r[1, :, 0])[0, 1]) < 5e-3 and abs(np.corrcoef(r[0, :, 0], r[0, :, 1])[0, 1]) < 5e-3
    assert not np.array_equal(noisy[0].cpu().numpy(), _synth(c, f1s[:1], None, bits[:1], nsamp, amp=amp, sigma=sigma, seed=100)[0].cpu().numpy())

    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"],
                           in_format=0, nstreams=B)
    maxf = h.max_frames_for(nsamp)

    def run(x):
        h.reset()
        ob = torch.zeros((B, maxf, h.Nbits), dtype=torch.uint8, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
        cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        h.demod_batch(x.data_ptr(), nsamp * 2, nsamp, ob.data_ptr(), maxf * h.Nbits, 0, 0, 0, 0,
                      nfr.data_ptr(), cons.data_ptr(), maxf, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return ob.cpu().numpy(), nfr.cpu().numpy()

    b0, n0 = run(clean)
    b1, n1 = run(noisy)
    nf = int(min(n0.min(), n1.min()))
    # both runs emit one continuous symbol stream; align each on the transmitted bits (fixed delay)
    def count(rx, tx):
        return min(int(np.count_nonzero(rx[200 + d:15200 + d] != tx[200:15200])) for d in range(0, 60))

    errs = tot = 0
    for s in range(B):
        assert count(b0[s, :nf].reshape(-1), bits[s]) == 0, ("clean stream not error free", s)
        errs += count(b1[s, :nf].reshape(-1), bits[s]); tot += 15000
    ber = errs / tot
    theory = 0.5 * np.exp(-ebno / 2.0)
    print(f"device AWGN Eb/N0 8 dB: BER {ber:.4f}, non-coherent 2FSK theory {theory:.4f}")
    assert 0.8 * theory < ber < 1.6 * theory, (ber, theory)


@pytest.mark.parametrize("D,n_in,byte_off,stride_pad", [
    (45, 45 * 700 + 79, 0, 0),        # several tiles, aligned
    (45, 45 * 300 + 100, 2, 6),       # 2-byte aligned but not 16: first-tile slow path + arithmetic conversion
    (45, 45 * 300 + 100, 1, 3),       # odd addresses: table conversion path
    (30, 30 * 1000 + 5, 0, 2),        # -a 80000 at 2.4 MS/s (README.md:172)
    (9, 9 * 3000 + 80, 4, 0),         # heavy overlap (L > 8 D)
    (7, 80, 0, 0),                    # exactly one output (n_in == padded tap count)
    (200, 200 * 300 + 79, 0, 10),     # window larger than one load group
    (45, 79, 0, 0),                   # shorter than the padded filter: no output
    (50, 50 * 600 + 90, 0, 0),        # csdr fir_decimate_cc 50 (README.md:162): the second shape of the convert-once kernel
    (50, 50 * 257 + 79, 2, 4),        # ... 2-byte aligned, a last tile of a few outputs
    (45, 45 * 1300 + 123, 0, 2),      # several tiles of 252 and a partial one
    (6, 6 * 5000 + 83, 0, 0),         # the systolic kernel's shapes (rtl_fsk's in-process decimations): 14 blocks per output
    (10, 10 * 3000 + 80, 2, 2),
    (18, 18 * 2000 + 99, 0, 4),
    (30, 30 * 997 + 79, 6, 0),
])
def test_decimator_shapes_alignment_and_batch(oracle, built_lib, D, n_in, byte_off, stride_pad):
    """fir_decimate_cc on the device against the oracle's scalar loop, bit for bit, for three streams laid
    out at awkward addresses (s16 and f32 outputs)."""
    import torch
    import pirip_amd
    L = oracle.lib()
    rng = np.random.default_rng(D * 1000 + byte_off)
    B = 3
    stride = 2 * n_in + stride_pad
    host = rng.integers(0, 256, byte_off + B * stride + 64, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    dec = pirip_amd.HipDecim(D, 0.05, out_s16=True)
    decf = pirip_amd.HipDecim(D, 0.05, out_s16=False)
    ntaps = L.oracle_firdes_filter_len(0.05)
    tp = np.zeros(80, dtype=np.float32)
    L.oracle_firdes_lowpass_f_hamming(tp.ctypes.data, ntaps, 0.5 / D)
    n_out = dec.nout(n_in)
    assert n_out == (0 if n_in < 80 else (n_in - 80) // D + 1)
    o16 = torch.full((B, max(n_out, 1), 2), 12345, dtype=torch.int16, device="cuda")
    o32 = torch.full((B, max(n_out, 1), 2), 7.0, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dec.batch(dev.data_ptr() + byte_off, stride, n_in, o16.data_ptr(), max(n_out, 1) * 4, B, st)
    decf.batch(dev.data_ptr() + byte_off, stride, n_in, o32.data_ptr(), max(n_out, 1) * 8, B, st)
    torch.cuda.synchronize()
    if n_out == 0:
        assert int((o16 != 12345).sum()) == 0 and int((o32 != 7.0).sum()) == 0
        return
    for s in range(B):
        u8 = np.ascontiguousarray(host[byte_off + s * stride: byte_off + s * stride + 2 * n_in]).reshape(n_in, 2)
        f = np.zeros(u8.shape, dtype=np.float32)
        L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
        y = np.zeros((n_in // D + 2, 2), dtype=np.float32)
        k = L.oracle_fir_decimate_cc(f.ctypes.data, y.ctypes.data, n_in, D, tp.ctypes.data, 80)
        assert k == n_out
        s16 = np.zeros((n_out, 2), dtype=np.int16)
        L.oracle_convert_f_s16(y.ctypes.data, s16.ctypes.data, 2 * n_out)
        assert np.array_equal(o32[s, :n_out].cpu().numpy().view(np.uint32), y[:n_out].view(np.uint32)), (s, "f32")
        assert np.array_equal(o16[s, :n_out].cpu().numpy(), s16), (s, "s16")


def test_general_kernel_odd_stream_counts_share_tables(oracle, built_lib, monkeypatch):
    """The general kernel packs several streams into one workgroup (shared tables in LDS): 1, 3 and 7 streams
    with different lengths of valid data, s16 input (config 3's demodulator side)."""
    import torch
    import pirip_amd
    monkeypatch.setenv("PIRIP_FORCE_GENERAL", "1")
    c = sigutil.CFG3
    for B in (1, 3, 7):
        xs = []
        for s in range(B):
            bits = np.random.default_rng(100 + s).integers(0, 2, 400).astype(np.uint8)
            x = sigutil.mod_complex(oracle, c, bits, f1=c["f1"] + 40 * s)
            xs.append(np.clip(np.rint(x[s * 3:] * 4000.0), -32768, 32767).astype(np.int16))
        nsamp = min(v.shape[0] for v in xs)
        host = np.stack([v[:nsamp] for v in xs])
        dev = torch.from_numpy(host).cuda()
        h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"],
                               in_format=pirip_amd.IN_CS16, nstreams=B)
        maxf = h.max_frames_for(nsamp)
        bits_d = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
        filt = torch.zeros((B, maxf, 100), dtype=torch.float32, device="cuda")
        stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        h.demod_batch(dev.data_ptr(), nsamp * 4, nsamp, bits_d.data_ptr(), maxf * 50, filt.data_ptr(), maxf * 100,
                      stats.data_ptr(), maxf * pirip_amd.STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), maxf, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for s in range(B):
            o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
            ro = o.demod(host[s], oracle.IN_CS16)
            n = int(nfr[s])
            rh = {"nframes": n, "consumed": int(cons[s]), "bits": bits_d[s, :n].cpu().numpy(),
                  "rx_filt": filt[s, :n].cpu().numpy(), "stats": stats[s, :n].cpu().numpy()}
            _compare(ro, rh)


def test_stream_scalars_after_a_call_without_stats_output(oracle, built_lib, kernel_choice):
    """The per-stream scalar state (tone estimates, timing, SNRest, nin, ppm) left behind by a launch that did not
    ask for per-frame stats equals the last frame's stats of a launch that did."""
    import torch
    import pirip_amd
    c = sigutil.CFG1
    u8, _ = sigutil.make_u8_stream(oracle, c, 4000, seed=5, ebno_db=9.0, random_bits=True)
    dev = torch.from_numpy(u8).cuda()
    nsamp = u8.shape[0]
    out = []
    for want_stats in (True, False):
        h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=0, nstreams=1)
        maxf = h.max_frames_for(nsamp)
        bits = torch.zeros((maxf, 50), dtype=torch.uint8, device="cuda")
        stats = torch.zeros((maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
        nfr = torch.zeros(1, dtype=torch.int32, device="cuda"); cons = torch.zeros(1, dtype=torch.int64, device="cuda")
        h.demod_batch(dev.data_ptr(), 0, nsamp, bits.data_ptr(), 0, 0, 0, stats.data_ptr() if want_stats else 0, 0,
                      nfr.data_ptr(), cons.data_ptr(), maxf, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sc = np.zeros(8, dtype=np.float32)
        assert h.L.pirip_hip_get_scalars(h.h, 0, sc.ctypes.data) == 0
        out.append((int(nfr[0]), sc.copy(), stats[int(nfr[0]) - 1].cpu().numpy(), bits.cpu().numpy()))
    (n1, sc1, st1, b1), (n2, sc2, _, b2) = out
    assert n1 == n2 and n1 >= 60 and np.array_equal(b1, b2)
    assert np.array_equal(sc1, sc2), (sc1, sc2)                      # same state with and without the stats output
    assert np.array_equal(sc1[[0, 1, 4, 5, 7]], st1[[0, 1, 4, 5, 7]])  # and it is the last frame's stats row


def test_two_handles_on_two_hip_streams_concurrently(oracle, built_lib):
    """Launches are asynchronous on the caller's HIP stream: two demodulators of different configurations (fast
    kernel / general kernel) and a decimator, enqueued on two streams without intermediate synchronisation, give
    what each gives alone."""
    import torch
    import pirip_amd
    c1, c3 = sigutil.CFG1, sigutil.CFG3
    B = 64
    u8, _ = sigutil.make_u8_stream(oracle, c1, 6000, seed=1, random_bits=True)
    x3 = sigutil.mod_complex(oracle, c3, np.random.default_rng(2).integers(0, 2, 600).astype(np.uint8))
    s16 = np.clip(np.rint(x3 * 5000.0), -32768, 32767).astype(np.int16)
    d1 = torch.from_numpy(u8).cuda(); d3 = torch.from_numpy(s16).cuda()
    n1, n3 = u8.shape[0], s16.shape[0]
    h1 = pirip_amd.HipDemod(c1["Fs"], c1["Rs"], 2, P=c1["P"], est_min=c1["est_min"], est_max=c1["est_max"], in_format=0, nstreams=B)
    h3 = pirip_amd.HipDemod(c3["Fs"], c3["Rs"], 2, P=c3["P"], est_min=c3["est_min"], est_max=c3["est_max"],
                            in_format=pirip_amd.IN_CS16, nstreams=B)
    m1, m3 = h1.max_frames_for(n1), h3.max_frames_for(n3)

    def outs(m):
        return (torch.zeros((B, m, 50), dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"),
                torch.zeros(B, dtype=torch.int64, device="cuda"))
    ref = []
    for h, d, n, m, es in ((h1, d1, n1, m1, 2), (h3, d3, n3, m3, 4)):          # alone, default stream
        b, f, c = outs(m)
        h.demod_batch(d.data_ptr(), 0, n, b.data_ptr(), m * 50, 0, 0, 0, 0, f.data_ptr(), c.data_ptr(), m, 0)
        torch.cuda.synchronize()
        ref.append((b.clone(), f.clone(), c.clone()))
        h.reset()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    b1, f1, k1 = outs(m1); b3, f3, k3 = outs(m3)
    for _ in range(3):                                                            # interleaved, no sync in between
        h1.reset(sa.cuda_stream); h3.reset(sb.cuda_stream)
        h1.demod_batch(d1.data_ptr(), 0, n1, b1.data_ptr(), m1 * 50, 0, 0, 0, 0, f1.data_ptr(), k1.data_ptr(), m1, sa.cuda_stream)
        h3.demod_batch(d3.data_ptr(), 0, n3, b3.data_ptr(), m3 * 50, 0, 0, 0, 0, f3.data_ptr(), k3.data_ptr(), m3, sb.cuda_stream)
    torch.cuda.synchronize()
    for got, want in (((b1, f1, k1), ref[0]), ((b3, f3, k3), ref[1])):
        assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and torch.equal(got[0], want[0])
    assert int(f1[0]) >= 100 and int(f3[0]) >= 10


def _random_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        Ts = int(rng.choice([8, 10, 12, 16, 20, 24, 32, 36, 40, 48, 60, 64, 80, 100]))
        divs = [p for p in range(4, Ts + 1) if Ts % p == 0]
        P = int(rng.choice(divs))
        Rs = int(rng.choice([100, 1200, 2400, 4800, 10000]))
        M = int(rng.choice([2, 4]))
        Fs = Ts * Rs
        k = int(rng.integers(1, 3))                      # tone spacing in multiples of Rs
        f1 = Rs * int(rng.integers(1, 3))
        if f1 + (M - 1) * k * Rs + Rs >= Fs // 2:
            continue
        out.append((Fs, Rs, M, P, f1, k * Rs, int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("Fs,Rs,M,P,f1,shift,fmt", _random_configs(12, seed=20260928))
def test_randomised_configurations_and_formats(oracle, built_lib, Fs, Rs, M, P, f1, shift, fmt):
    """Twelve configurations drawn from everything fsk_create_hbr accepts (Ts 8..100, every legal P, 2-/4-FSK),
    each through one of the three input formats, timing offset and tone offset randomised, clean signal:
    bits, tone estimates and the nin sequence exact, soft magnitudes within tolerance."""
    import pirip_amd
    rng = np.random.default_rng(Fs * 7 + P)
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1 + int(rng.integers(-Rs // 4, Rs // 4)), shift=shift,
             est_min=Rs // 2, est_max=min(Fs // 2 - Rs, f1 + M * shift + 2 * Rs))
    bits = rng.integers(0, 2, 4000 * (1 if M == 2 else 2)).astype(np.uint8)
    x = sigutil.mod_complex(oracle, c, bits)[int(rng.integers(0, Fs // Rs)):]
    if fmt == 0:
        buf = oracle.quantise_cu8(x, amp=25.0); fo, fh = oracle.IN_CU8_FSKDEMOD, 0
    elif fmt == 1:
        buf = np.clip(np.rint(x * 6000.0), -32768, 32767).astype(np.int16); fo, fh = oracle.IN_CS16, pirip_amd.IN_CS16
    else:
        buf = np.ascontiguousarray(x, dtype=np.float32); fo, fh = oracle.IN_CF32, pirip_amd.IN_CF32
    o = oracle.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"])
    h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], in_format=fh, nstreams=1)
    ro = o.demod(buf, fo); rh = h.demod_host(buf)
    assert ro["nframes"] >= 30
    _compare(ro, rh)


@pytest.mark.parametrize("ebno_db,dc,amp", [(1.0, 0.0, 10.0), (3.0, 12.0, 14.0), (12.0, 0.0, 70.0)])
def test_stress_low_snr_dc_offset_and_clipping(oracle, built_lib, kernel_choice, ebno_db, dc, amp):
    """Inputs that push the estimators around: Eb/N0 of 1-3 dB (the spectral peaks wander, blanking and the
    nin feedback are exercised every few frames), a DC offset on the u8 samples, and an amplitude that clips the
    8-bit range. Tone estimates, frame counts and nin still have to match exactly; bits up to near-tie flips."""
    c = sigutil.CFG1
    rng = np.random.default_rng(int(ebno_db * 10) + 7)
    bits = rng.integers(0, 2, 40000).astype(np.uint8)
    x = sigutil.add_awgn(sigutil.mod_complex(oracle, c, bits)[5:], ebno_db, c, rng)
    u8 = np.clip(np.rint(127.0 + dc + amp * x.astype(np.float64)), 0, 255).astype(np.uint8)
    o, h = _pair(oracle, c, 0, 0)
    ro = o.demod(u8, oracle.IN_CU8_FSKDEMOD)
    rh = h.demod_host(u8)
    assert ro["nframes"] >= 790
    nflips = _compare(ro, rh, allow_near_tie_flips=True)
    nin_changes = int(np.count_nonzero(np.diff(ro["stats"][:, 6])))
    fest_changes = int(np.count_nonzero(np.diff(ro["stats"][:, 0])) + np.count_nonzero(np.diff(ro["stats"][:, 1])))
    print(f"Eb/N0 {ebno_db} dB dc {dc} amp {amp}: {nflips} near-tie flips, nin changed {nin_changes}x, f_est changed {fest_changes}x")
    assert nflips <= 8


@pytest.mark.parametrize("M,mask", [(2, 0), (2, 2000), (4, 0), (4, 2000)])
def test_block_instance_serves_rtl_fsk_r1000_and_carries_state(oracle, built_lib, M, mask):
    """`rtl_fsk -r 1000` at 240 kS/s (README.md:152,184; -m 4 --mask 2000: README.md:239): Ts = 240, Ndft = 4096 runs on the
    workgroup-per-stream block instance (fsk_demod_block.hip). One shot, ragged chunks (the raw tail / last tone estimates / Sf
    carried between launches), a batch of streams at sample-aligned strides and a max_frames stop all equal the oracle."""
    import torch
    import pirip_amd
    Fs, Rs, P = 240000, 1000, 15
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=11000, shift=2000, est_min=500, est_max=Fs // 2 - Rs)
    rng = np.random.default_rng(100 + M + mask)
    tol = RX_FILT_TOL * 5.0                                         # scaled by N / 2400
    mk = lambda: oracle.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], tone_spacing=mask if mask else 100, mask=bool(mask))
    streams = []
    for s in range(3):
        bits = rng.integers(0, 2, 2500 * (1 if M == 2 else 2)).astype(np.uint8)
        x = sigutil.mod_complex(oracle, c, bits)[int(rng.integers(0, 240)):]
        if s == 1:
            x = sigutil.add_awgn(x, 9.0, c, rng)
        streams.append(oracle.quantise_cu8(x, amp=20.0))
    # one stream: one shot and ragged chunks
    u8 = streams[1]
    ro = mk().demod(u8, oracle.IN_CU8_CSDR)
    h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=1)
    assert h.kernel() == "block" and "fsk_demod_block_kernel" in h.kernel_name()
    _compare(ro, h.demod_host(u8), tol=tol, allow_near_tie_flips=True, M=M)
    # nothing, one sample short of a frame, exactly one frame, a saturated input
    he = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=1)
    r0 = he.demod_host(np.zeros((0, 2), dtype=np.uint8))
    assert r0["nframes"] == 0 and r0["consumed"] == 0
    r0 = he.demod_host(u8[:11999])
    assert r0["nframes"] == 0 and r0["consumed"] == 0
    r1 = he.demod_host(u8[:12000])
    o1 = mk().demod(u8[:12000], oracle.IN_CU8_CSDR)
    assert r1["nframes"] == 1 == o1["nframes"] and r1["consumed"] == 12000 and np.array_equal(r1["bits"], o1["bits"])
    hz = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=1)
    z = np.full((36500, 2), 255, dtype=np.uint8)
    rz, oz = hz.demod_host(z), mk().demod(z, oracle.IN_CU8_CSDR)
    assert rz["nframes"] == oz["nframes"] == 3 and np.array_equal(rz["bits"], oz["bits"]) and np.array_equal(rz["stats"][:, :4], oz["stats"][:, :4])
    h2 = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=1)
    pos, carry = 0, np.zeros((0, 2), dtype=np.uint8)
    bits_l, filt_l, stats_l = [], [], []
    while pos < u8.shape[0]:
        n = int(rng.integers(1, 40000))
        buf = np.concatenate([carry, u8[pos:pos + n]]); pos += n
        r = h2.demod_host(buf)
        bits_l.append(r["bits"]); filt_l.append(r["rx_filt"]); stats_l.append(r["stats"])
        carry = buf[r["consumed"]:]
    rh = {"nframes": sum(len(b) for b in bits_l), "consumed": u8.shape[0] - carry.shape[0], "bits": np.concatenate(bits_l),
          "rx_filt": np.concatenate(filt_l), "stats": np.concatenate(stats_l)}
    _compare(ro, rh, tol=tol, allow_near_tie_flips=True, M=M)
    # a batch: three streams at a stride that is only sample-aligned, stopped after 7 frames and resumed
    nsamp = min(x.shape[0] for x in streams)
    stride = nsamp * 2 + 6
    flat = np.zeros(3 * stride + 64, dtype=np.uint8)
    for s in range(3):
        flat[2 + s * stride: 2 + s * stride + 2 * nsamp] = streams[s][:nsamp].reshape(-1)
    dev = torch.from_numpy(flat).cuda()
    hb = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=3)
    maxf = hb.max_frames_for(nsamp)
    bits = torch.zeros((3, maxf, hb.Nbits), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((3, maxf, M * 50), dtype=torch.float32, device="cuda")
    stats = torch.zeros((3, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(3, dtype=torch.int32, device="cuda")
    cons = torch.zeros(3, dtype=torch.int64, device="cuda")
    hb.demod_batch(dev.data_ptr() + 2, stride, nsamp, bits.data_ptr(), maxf * hb.Nbits, filt.data_ptr(), maxf * M * 50, stats.data_ptr(),
                   maxf * pirip_amd.STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), 7, 0)
    torch.cuda.synchronize()
    assert list(nfr.cpu().numpy()) == [7, 7, 7]
    first = [(bits[s, :7].cpu().numpy(), filt[s, :7].cpu().numpy(), stats[s, :7].cpu().numpy(), int(cons[s])) for s in range(3)]
    for s in range(3):
        # resume every stream from where it stopped (per-stream consumed counts differ): one launch per stream offset
        h1 = None
    # second call: present each stream's unconsumed tail at the front of a fresh buffer
    L = min(nsamp - f[3] for f in first)
    flat2 = np.zeros(3 * (2 * L + 10), dtype=np.uint8)
    for s in range(3):
        flat2[s * (2 * L + 10): s * (2 * L + 10) + 2 * L] = streams[s][first[s][3]: first[s][3] + L].reshape(-1)
    dev2 = torch.from_numpy(flat2).cuda()
    hb.demod_batch(dev2.data_ptr(), 2 * L + 10, L, bits.data_ptr(), maxf * hb.Nbits, filt.data_ptr(), maxf * M * 50, stats.data_ptr(),
                   maxf * pirip_amd.STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), maxf, 0)
    torch.cuda.synchronize()
    for s in range(3):
        n2 = int(nfr[s])
        rh = {"nframes": 7 + n2, "consumed": first[s][3] + int(cons[s]), "bits": np.concatenate([first[s][0], bits[s, :n2].cpu().numpy()]),
              "rx_filt": np.concatenate([first[s][1], filt[s, :n2].cpu().numpy()]), "stats": np.concatenate([first[s][2], stats[s, :n2].cpu().numpy()])}
        ro_s = mk().demod(streams[s][:first[s][3] + L], oracle.IN_CU8_CSDR)
        _compare(ro_s, rh, tol=tol, allow_near_tie_flips=(s == 1), M=M)


# ---- the first frame after create / reset in the oracle's own operation order ------------------------------------
def test_device_atan2f_restatement_equals_the_hosts_libm(built_lib):
    """The exact first frame's timing angle is glibc's atan2f restated in device code (fdlibm's float algorithm): bit for bit the host
    libm's atan2f on 4 * 10^6 arguments of every quadrant, scale and the special values."""
    import ctypes as C
    import torch
    import pirip_amd
    libm = C.CDLL("libm.so.6")
    rng = np.random.default_rng(3)
    n = 4_000_000
    y = (rng.standard_normal(n) * rng.choice([1e-6, 1e-3, 1.0, 1e3, 1e9], n)).astype(np.float32)
    x = (rng.standard_normal(n) * rng.choice([1e-6, 1e-3, 1.0, 1e3, 1e9], n)).astype(np.float32)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-45, -1e-45, 3e38, 1e-38], dtype=np.float32)
    y[:100] = np.repeat(sp, 10); x[:100] = np.tile(sp, 10)
    want = np.zeros(n, dtype=np.float32)
    # host atan2f through a tiny vectorising helper: ctypes call per element is too slow for 4e6, so compile nothing -- use numpy's
    # frompyfunc on chunks of the special values only, and the C library's vector-free loop via a shared buffer for the rest
    libm.atan2f.restype = C.c_float; libm.atan2f.argtypes = [C.c_float, C.c_float]
    idx = np.concatenate([np.arange(100), rng.choice(n, 200000, replace=False)])
    for i in idx:
        want[i] = libm.atan2f(float(y[i]), float(x[i]))
    dy, dx = torch.from_numpy(y).cuda(), torch.from_numpy(x).cuda()
    out = torch.zeros(n, dtype=torch.float32, device="cuda")
    L = pirip_amd.lib()
    L.pirip_hip_selftest_atan2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    assert L.pirip_hip_selftest_atan2(dy.data_ptr(), dx.data_ptr(), out.data_ptr(), n) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[idx].view(np.uint32), want[idx].view(np.uint32)), np.nonzero(got[idx].view(np.uint32) != want[idx].view(np.uint32))[0][:5]


@pytest.mark.parametrize("kernel", ["wave", "general"])
def test_first_frame_is_the_oracles_bit_for_bit_at_every_start_offset(oracle, built_lib, kernel, monkeypatch):
    """SURVEY.md 8d config 2 prescribes per-stream start offsets; at `-p 24` (P == Ts) a recording that starts one sample before a symbol
    boundary hands the first decision ONE sample: both tone magnitudes are then equal up to float rounding (margin 1e-10 of the peak)
    and the decision follows the last bits of the timing estimate. The prologue kernel (fsk_demod_exact0_kernel) performs the oracle's
    operations in its order, so frame 0 -- bits, soft magnitudes, timing, SNR terms -- is bit for bit the oracle's on the wave-kernel
    handle and on the any-configuration one, for every offset x tone plan; every later bit of the noise-free stream equals too."""
    import torch
    import pirip_amd
    if kernel == "general":
        monkeypatch.setenv("PIRIP_KERNEL", "general")
    c = sigutil.CFG1
    streams, refs = [], []
    for plan in range(5):
        for off in range(24):
            u8, _ = sigutil.make_u8_stream(oracle, c, 1000, offset=off, tone_bins=plan - 2)
            streams.append(u8[:19 * 1200 + 30])
    n = min(len(s) for s in streams)
    host = np.stack([s[:n] for s in streams])
    B = host.shape[0]
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], nstreams=B)
    assert h.kernel() == kernel
    dev = torch.from_numpy(host).cuda()
    maxf = h.max_frames_for(n)
    bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, maxf, 100), dtype=torch.float32, device="cuda")
    stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # two calls: the first holds each stream's first frames (prologue + demodulator proper), the second carries on (no prologue)
    n1 = 5 * 1200 + 77
    h.demod_batch(dev.data_ptr(), n * 2, n1, bits.data_ptr(), maxf * 50, filt.data_ptr(), maxf * 100, stats.data_ptr(), maxf * pirip_amd.STATS_PER_FRAME,
                  nfr.data_ptr(), cons.data_ptr(), maxf, st)
    torch.cuda.synchronize()
    nf1, c1 = nfr.cpu().numpy().copy(), cons.cpu().numpy().copy()
    hb, hf, hs = bits.cpu().numpy().copy(), filt.cpu().numpy().copy(), stats.cpu().numpy().copy()
    ntie = 0
    for s in range(B):
        o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        ro = o.demod(host[s, :n1], oracle.IN_CU8_FSKDEMOD)
        assert ro["nframes"] == nf1[s] >= 4 and ro["consumed"] == c1[s]
        f0 = ro["rx_filt"].reshape(-1, 2, 50)[0]
        assert np.array_equal(hb[s, 0], ro["bits"][0]), (s, "first frame's bits")
        assert np.array_equal(hf[s, 0].view(np.uint32), f0.reshape(-1).view(np.uint32)), (s, "first frame's soft magnitudes")
        assert np.array_equal(hs[s, 0, :5].view(np.uint32), ro["stats"][0, :5].view(np.uint32)), (s, "tone estimates / timing of the first frame")
        assert np.array_equal(hs[s, 0, 6], ro["stats"][0, 6])
        assert np.array_equal(hb[s, :nf1[s]], ro["bits"]), (s, "later frames")
        ntie += int(abs(f0[0, 0] - f0[1, 0]) < 1e-6 * np.abs(f0).max() and f0[0, 0] > 0)
    assert ntie >= 4            # the offsets that make the first decision a rounding tie are in the set (one per tone plan)
    h.close()


@pytest.mark.parametrize("case", ["2fsk_p24_u8", "4fsk_p8_u8", "2fsk_1k_s16", "4fsk_mask_f32"])
def test_exact_kernel_equals_the_oracle_bit_for_bit_under_noise(oracle, built_lib, case, monkeypatch):
    """PIRIP_KERNEL=exact (fsk_demod_exact_kernel: every frame in the oracle's operation order -- serial oscillator recursion carried
    across frames, forward window sums, serial timing and power sums, glibc's atan2f, the double-precision ppm smoothing) reproduces the
    oracle under NOISE, where the fast kernels differ from it at near-ties: every bit, every soft magnitude, every per-frame statistic
    (tone estimates, timing, SNRest, nin, ppm, signal and noise power) identical word for word, at Eb/N0 from 0 to 9 dB, with start
    offsets, detuned tone plans and calls split at arbitrary sample counts. This is the device-side proof that what separates the fast
    kernels from the oracle under noise is summation order only: same inputs, same algorithm, the oracle's order -> zero differing words."""
    import torch
    import pirip_amd
    monkeypatch.setenv("PIRIP_KERNEL", "exact")
    mask = 0
    if case == "2fsk_p24_u8":
        c, fmt, ofmt = sigutil.CFG1, pirip_amd.IN_CU8_FSKDEMOD, oracle.IN_CU8_FSKDEMOD
    elif case == "4fsk_p8_u8":
        c, fmt, ofmt = sigutil.CFG4, pirip_amd.IN_CU8_FSKDEMOD, oracle.IN_CU8_FSKDEMOD
    elif case == "2fsk_1k_s16":
        c, fmt, ofmt = sigutil.CFG3, pirip_amd.IN_CS16, oracle.IN_CS16
    else:
        c, fmt, ofmt, mask = sigutil.CFG4, pirip_amd.IN_CF32, oracle.IN_CF32, 10000
    Ts = c["Fs"] // c["Rs"]
    nbits = 20 * 50 * (2 if c["M"] == 4 else 1)
    streams = []
    rng = np.random.default_rng(11)
    for k, ebno in enumerate([0.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.5, 9.0, None]):
        off = int(rng.integers(0, Ts))
        x = sigutil.mod_complex(oracle, c, rng.integers(0, 2, nbits).astype(np.uint8), f1=c["f1"] + int(rng.integers(-300, 300)))
        if ebno is not None:
            x = sigutil.add_awgn(x, ebno, c, rng)
        if fmt == pirip_amd.IN_CF32:
            buf = np.ascontiguousarray(x[off:]).astype(np.float32)
        elif fmt == pirip_amd.IN_CS16:
            buf = np.ascontiguousarray(np.clip(np.round(x[off:] * 4000.0), -32768, 32767).astype(np.int16))
        else:
            buf = np.ascontiguousarray(oracle.quantise_cu8(x, amp=32.0)[off:])
        streams.append(buf)
    n = min(len(s) for s in streams)
    host = np.stack([s[:n] for s in streams])
    B = host.shape[0]
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=fmt, nstreams=B)
    assert h.kernel() == "exact" and "exact" in h.kernel_name()
    dev = torch.from_numpy(host).cuda()
    bps = host.dtype.itemsize * 2
    nb, nf_ = h.Nbits, c["M"] * 50
    maxf = h.max_frames_for(n)
    st = torch.cuda.current_stream().cuda_stream
    got_b = [[] for _ in range(B)]; got_f = [[] for _ in range(B)]; got_s = [[] for _ in range(B)]
    # the recording in three calls cut at odd sample counts: a stream's unconsumed tail is presented again at the head of the next call
    done = np.zeros(B, dtype=np.int64)
    cuts = [n // 3 + 17, (2 * n) // 3 + 5, n]
    for cut in cuts:
        # every stream resumes at its own consumed position: one launch per stream set sharing a position is overkill here, so streams
        # are re-packed host-side at their positions (the device call sees B equal-length buffers)
        ln = int(min(cut - done.max(), n - done.max()))
        if ln <= 0:
            continue
        pack = np.stack([host[s, done[s]:done[s] + ln] for s in range(B)])
        dpk = torch.from_numpy(np.ascontiguousarray(pack)).cuda()
        bits = torch.zeros((B, maxf, nb), dtype=torch.uint8, device="cuda")
        filt = torch.zeros((B, maxf, nf_), dtype=torch.float32, device="cuda")
        stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        h.demod_batch(dpk.data_ptr(), ln * bps, ln, bits.data_ptr(), maxf * nb, filt.data_ptr(), maxf * nf_, stats.data_ptr(),
                      maxf * pirip_amd.STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), maxf, st)
        torch.cuda.synchronize()
        nfh, ch = nfr.cpu().numpy(), cons.cpu().numpy()
        for s in range(B):
            got_b[s].append(bits[s, :nfh[s]].cpu().numpy()); got_f[s].append(filt[s, :nfh[s]].cpu().numpy()); got_s[s].append(stats[s, :nfh[s]].cpu().numpy())
        done += ch
    del dev
    words = 0
    for s in range(B):
        o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], mask=bool(mask),
                             tone_spacing=mask if mask else 100)
        ro = o.demod(host[s, :done[s]], ofmt)                # the oracle in ONE call on the samples the three device calls consumed
        b = np.concatenate(got_b[s]); f = np.concatenate(got_f[s]); sv = np.concatenate(got_s[s])
        assert ro["nframes"] == len(b) >= 14 and ro["consumed"] == done[s], (s, ro["nframes"], len(b), ro["consumed"], done[s])
        assert np.array_equal(b, ro["bits"]), (s, "bits", int((b != ro["bits"]).sum()))
        assert np.array_equal(f.view(np.uint32), ro["rx_filt"].view(np.uint32)), (s, "soft magnitudes", int((f.view(np.uint32) != ro["rx_filt"].view(np.uint32)).sum()))
        assert np.array_equal(sv[:, :10].view(np.uint32), ro["stats"][:, :10].view(np.uint32)), \
            (s, "statistics", np.argwhere(sv[:, :10].view(np.uint32) != ro["stats"][:, :10].view(np.uint32))[:6].tolist())
        words += b.size + f.size + sv[:, :10].size
    assert words > 20000
    h.close()


@pytest.mark.parametrize("fmt", ["fskdemod", "csdr"])
def test_band_only_estimator_changes_no_output(oracle, built_lib, fmt):
    """pirip_hip_set_estimator_band_only (opt-in): the estimator computes and smooths only the FFT bins the peak search can read.
    Against a handle with the full estimator, on noisy streams whose tones sit at the low edge, in the middle and at the high
    edge of the search range, handed over in uneven pieces: every output word (bits, soft magnitudes, all stats columns,
    frame / sample counts) is IDENTICAL, Sf inside the band is bit-identical (and equal to the oracle's), Sf outside the band is
    not maintained; a search range that leaves the band is refused, one inside it is honoured like the full estimator honours it."""
    import ctypes as C
    import pirip_amd
    c = dict(sigutil.CFG1)
    fo, fh = (oracle.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_FSKDEMOD) if fmt == "fskdemod" else (oracle.IN_CU8_CSDR, pirip_amd.IN_CU8_CSDR)
    for k, (f1, shift, ebno, seed) in enumerate([(1500, 9000, 9.0, 1), (10000, 10000, 6.0, 2), (14000, 10400, 4.0, 3), (10000, 10000, None, 4)]):
        cc = dict(c, f1=f1, shift=shift)
        u8, _ = sigutil.make_u8_stream(oracle, cc, 30000, seed=seed, offset=3 + 5 * k, ebno_db=ebno, random_bits=True)
        if fmt == "csdr":
            u8 = np.ascontiguousarray(np.clip(u8.astype(np.int32) + 1, 0, 255).astype(np.uint8))
        full = pirip_amd.HipDemod(c["Fs"], c["Rs"], 2, P=24, est_min=c["est_min"], est_max=c["est_max"], in_format=fh, nstreams=1)
        band = pirip_amd.HipDemod(c["Fs"], c["Rs"], 2, P=24, est_min=c["est_min"], est_max=c["est_max"], in_format=fh, nstreams=1)
        band.set_estimator_band_only(True)
        assert "band-only" in band.kernel_name() and "band-only" not in full.kernel_name()
        rf, rb = [], []
        pos = 0
        for n in (5000, 1234, 40000, 7, 10 ** 9):
            piece = u8[pos:pos + n]
            pos += len(piece)
            if not len(piece):
                break
            rf.append(full.demod_host(piece)); rb.append(band.demod_host(piece))
            pos -= len(piece) - rf[-1]["consumed"]                 # the unconsumed tail goes in front of the next piece
            assert rb[-1]["consumed"] == rf[-1]["consumed"] and rb[-1]["nframes"] == rf[-1]["nframes"]
        for a, b in zip(rf, rb):
            assert np.array_equal(a["bits"], b["bits"])
            assert np.array_equal(a["rx_filt"].view(np.uint32), b["rx_filt"].view(np.uint32))
            assert np.array_equal(a["stats"].view(np.uint32), b["stats"].view(np.uint32))
        Sf_f, Sf_b = full.get_Sf(0), band.get_Sf(0)
        assert np.array_equal(Sf_f[128:160].view(np.uint32), Sf_b[128:160].view(np.uint32))
        assert not np.array_equal(Sf_b[160:], Sf_f[160:])           # (not maintained: what the stream's exact first frame left there)
        if k == 1:
            o = oracle.OracleFsk(c["Fs"], c["Rs"], 2, P=24, est_min=c["est_min"], est_max=c["est_max"])
            ro = o.demod(u8, fo)
            Sf_o = np.ctypeslib.as_array(C.cast(_oracle_field_Sf(oracle, o), C.POINTER(C.c_float)), shape=(256,)).copy()
            assert ro["nframes"] == sum(r["nframes"] for r in rb)
            assert np.array_equal(Sf_b[128:160], Sf_o[128:160])
            assert np.array_equal(np.concatenate([r["stats"][:, :4] for r in rb]), ro["stats"][:, :4])
        # the search range on a live handle: inside the band it moves (like the full estimator's), out of it it is refused
        assert band.set_freq_est_limits(500, 40000) != 0 and band.set_freq_est_limits(-5000, 20000) != 0
        assert band.set_freq_est_limits(2000, 28000) == 0 and full.set_freq_est_limits(2000, 28000) == 0
        more, _ = sigutil.make_u8_stream(oracle, cc, 6000, seed=seed + 10, offset=1, ebno_db=ebno, random_bits=True)
        a, b = full.demod_host(more), band.demod_host(more)
        assert np.array_equal(a["bits"], b["bits"]) and np.array_equal(a["stats"].view(np.uint32), b["stats"].view(np.uint32))


def test_band_only_estimator_4fsk_64_bin_band(oracle, built_lib):
    """The 4-FSK P = 8 instances carry a 64-bin band (search range 500 .. 60000 Hz, tools/bench_configs.py's config 4): bits, soft
    magnitudes and stats identical to the full estimator's under noise and across calls, Sf[128:192] bit-identical."""
    import pirip_amd
    c = dict(sigutil.CFG4, P=8)
    u8, _ = sigutil.make_u8_stream(oracle, c, 40000, seed=9, offset=11, ebno_db=7.0, random_bits=True)
    full = pirip_amd.HipDemod(c["Fs"], c["Rs"], 4, P=8, est_min=500, est_max=60000, nstreams=1)
    band = pirip_amd.HipDemod(c["Fs"], c["Rs"], 4, P=8, est_min=500, est_max=60000, nstreams=1)
    band.set_estimator_band_only(True)
    assert "band-only" in band.kernel_name()
    pos = 0
    for n in (7000, 301, 10 ** 9):
        a, b = full.demod_host(u8[pos:pos + n]), band.demod_host(u8[pos:pos + n])
        assert a["nframes"] == b["nframes"] > 0 or n == 301
        assert np.array_equal(a["bits"], b["bits"])
        assert np.array_equal(a["rx_filt"].view(np.uint32), b["rx_filt"].view(np.uint32))
        assert np.array_equal(a["stats"].view(np.uint32), b["stats"].view(np.uint32))
        pos += a["consumed"]
    Sf_f, Sf_b = full.get_Sf(0), band.get_Sf(0)
    assert np.array_equal(Sf_f[128:192].view(np.uint32), Sf_b[128:192].view(np.uint32)) and not np.array_equal(Sf_f[192:], Sf_b[192:])
    assert band.set_freq_est_limits(500, 61000) != 0 and band.set_freq_est_limits(1000, 59000) == 0


def test_band_only_estimator_is_refused_where_it_does_not_apply(oracle, built_lib):
    import pirip_amd
    c = sigutil.CFG1
    wide = pirip_amd.HipDemod(c["Fs"], c["Rs"], 2, P=24, est_min=500, est_max=40000, nstreams=1)          # the search reads bins beyond 31
    with pytest.raises(Exception):
        wide.set_estimator_band_only(True)
    c4 = sigutil.CFG4
    four = pirip_amd.HipDemod(c4["Fs"], c4["Rs"], 4, P=6, est_min=500, est_max=25000, nstreams=1)          # no instance was built for this shape
    with pytest.raises(Exception):
        four.set_estimator_band_only(True)
    neg = pirip_amd.HipDemod(c["Fs"], c["Rs"], 2, P=24, est_min=-20000, est_max=25000, nstreams=1)        # negative frequencies in the range
    with pytest.raises(Exception):
        neg.set_estimator_band_only(True)
    r = neg.demod_host(sigutil.make_u8_stream(oracle, c, 3000)[0])
    assert r["nframes"] > 0                                                                                  # ... and the handle is unharmed



# ---- the recalled constants as plan data (include/pirip_hip.h: pirip_fsk_recalled; oracle/fsk_oracle.h: fsk_oracle_recalled) ------------
_TABLE_ONLY = {"hann_denominator_ndft", "tc", "est_space_rs"}        # table / plan data for every kernel: the specialised instance stays


@pytest.mark.parametrize("field", sorted(__import__("oracle.binding", fromlist=["x"]).RECALLED_ALTERNATIVES))
def test_recalled_constant_at_its_alternative_value_hip_equals_oracle(oracle, built_lib, field):
    """Pin-day drill: every constant this repository holds from recall of codec2 is ONE field of the plan, in the product and in the
    CPU restatement alike. Flip one field to its plausible other value in both: the HIP path still equals the oracle under the
    same contract as at the defaults (noisy input, so that the estimator's details matter) -- so the day oracle/_ref exists and
    tests/golden/PINNED.json names a differing case, the repair is a default, not an edit of a kernel. Fields the specialised
    kernels are built around send the handle to the any-configuration kernel; table-only fields keep the wave instance."""
    import pirip_amd
    alt = oracle.RECALLED_ALTERNATIVES[field]
    c = sigutil.CFG1
    s16 = field == "s16_scale"
    if s16:
        c = dict(sigutil.CFG3)
        x = sigutil.mod_complex(oracle, c, np.random.default_rng(3).integers(0, 2, 6000).astype(np.uint8))
        x = x + np.random.default_rng(4).normal(0.0, 0.35, x.shape).astype(np.float32)
        buf = np.clip(np.rint(x * 6000.0), -32768, 32767).astype(np.int16)
        fmt_o, fmt_h = oracle.IN_CS16, pirip_amd.IN_CS16
    else:
        buf, _ = sigutil.make_u8_stream(oracle, c, 60000, seed=11, ebno_db=9.0, random_bits=True, amp=18.0, offset=5)
        fmt_o, fmt_h = oracle.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_FSKDEMOD
    kw = dict(P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    o = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], recalled={field: alt}, **kw)
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], in_format=fmt_h, recalled={field: alt}, **kw)
    o0 = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], **kw)
    h0 = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], in_format=fmt_h, **kw)
    assert h0.kernel() == "wave"
    assert h.kernel() == ("wave" if field in _TABLE_ONLY else "general"), (field, h.kernel())
    ro, rh = o.demod(buf, fmt_o), h.demod_host(buf)
    nflips = _compare(ro, rh, allow_near_tie_flips=True)
    # the field does something: the oracle at the alternative differs from the oracle at the default somewhere it can be seen
    r0 = o0.demod(buf, fmt_o)
    n = min(r0["nframes"], ro["nframes"])
    Sf_a = np.ctypeslib.as_array(__import__("ctypes").cast(_oracle_field_Sf(oracle, o), __import__("ctypes").POINTER(__import__("ctypes").c_float)), shape=(h.info.Ndft,)).copy()
    Sf_0 = np.ctypeslib.as_array(__import__("ctypes").cast(_oracle_field_Sf(oracle, o0), __import__("ctypes").POINTER(__import__("ctypes").c_float)), shape=(h0.info.Ndft,)).copy()
    differs = (ro["nframes"] != r0["nframes"] or Sf_a.shape != Sf_0.shape or not np.array_equal(Sf_a, Sf_0) or
               not np.array_equal(ro["rx_filt"][:n], r0["rx_filt"][:n]) or not np.array_equal(ro["stats"][:n], r0["stats"][:n]))
    assert differs, field
    assert np.array_equal(h.get_Sf(0), Sf_a), "smoothed spectrum differs at the alternative value"
    print(f"{field} = {alt}: {ro['nframes']} frames, kernel {h.kernel()}, {nflips} near-tie flips; default handle stays on {h0.kernel()}")


@pytest.mark.parametrize("shape", [
    # Fs, Rs, M, P (= Ts), mask spacing: the other P == Ts wave instances behind the exact first-frame prologue (rtl_fsk -a 80000 / 100000 -r 10000)
    (80000, 10000, 2, 8, 0), (80000, 10000, 4, 8, 8000), (100000, 10000, 2, 10, 0), (100000, 10000, 4, 10, 10000),
], ids=lambda s: "Fs%d-M%d-P%d-mask%d" % (s[0], s[2], s[3], s[4]))
def test_first_frame_is_bit_for_bit_on_the_cf32_p_equals_ts_instances(oracle, built_lib, shape):
    """The exact first-frame prologue writes the wave kernel's private state block (raw tail, per-tone phase step and table row, nin) from
    the general kernel's body for EVERY P == Ts wave instance -- also the complex-float Ts = 8 / 10 ones, 2- and 4-FSK, peak and mask
    estimator (the FreeDV shim's default shape is among them). Noise-free streams at several start offsets: frame 0 is the oracle's bit
    for bit (bits, soft magnitudes, tone estimates, timing, next nin), the frames behind it -- which start from the state block the
    prologue left -- meet the usual contract, in one call and split across two; with the prologue switched off the call still
    demodulates (pirip_hip_set_exact_first_frame, the A/B switch of binding.py)."""
    import pirip_amd
    Fs, Rs, M, P, mask = shape
    Ts = Fs // Rs
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=mask if mask else 10000, shift=mask if mask else 10000)
    bits = np.random.default_rng(60 + P + M).integers(0, 2, 50 * 12 * (M // 2)).astype(np.uint8)
    x = sigutil.mod_complex(oracle, c, bits) * np.float32(0.4)
    est_max = Fs // 2 - Rs
    for off in (0, 1, Ts // 2, Ts - 1):
        buf = np.ascontiguousarray(x[off:])
        o = oracle.OracleFsk(Fs, Rs, M, P=P, est_min=Rs // 2, est_max=est_max, mask=bool(mask), tone_spacing=mask if mask else 100)
        ro = o.demod(buf, oracle.IN_CF32)
        h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=Rs // 2, est_max=est_max, mask=mask, in_format=pirip_amd.IN_CF32)
        assert h.kernel() == "wave", (shape, h.kernel_name())
        rh = h.demod_host(buf)
        assert rh["nframes"] == ro["nframes"] >= 8
        assert np.array_equal(rh["bits"][0], ro["bits"][0]), (shape, off, "first frame's bits")
        assert np.array_equal(rh["rx_filt"][0].view(np.uint32), ro["rx_filt"][0].view(np.uint32)), (shape, off, "first frame's soft magnitudes")
        assert np.array_equal(rh["stats"][0, :5].view(np.uint32), ro["stats"][0, :5].view(np.uint32)), (shape, off, "tone estimates / timing of the first frame")
        assert rh["stats"][0, 6] == ro["stats"][0, 6]
        _compare(ro, rh, M=M)
        # the same stream in two calls: the second starts from the state the first left (no prologue then)
        h2 = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=Rs // 2, est_max=est_max, mask=mask, in_format=pirip_amd.IN_CF32)
        n1 = 3 * Ts * 50 + 7
        r1 = h2.demod_host(buf[:n1])
        r2 = h2.demod_host(buf[r1["consumed"]:])
        assert r1["nframes"] + r2["nframes"] == ro["nframes"]
        assert np.array_equal(np.concatenate([r1["bits"], r2["bits"]]), ro["bits"]), (shape, off, "split calls")
        # prologue off: frame 0 comes from the wave kernel itself (tolerance contract)
        h3 = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=Rs // 2, est_max=est_max, mask=mask, in_format=pirip_amd.IN_CF32)
        h3.set_exact_first_frame(False)
        r3 = h3.demod_host(buf)
        assert r3["nframes"] == ro["nframes"] and sigutil.rel_err(r3["rx_filt"], ro["rx_filt"]) < 3 * RX_FILT_TOL


@pytest.mark.parametrize("D", [45, 50, 6, 9, 10, 18, 30])
def test_convert_once_decimator_equals_the_per_output_one(oracle, built_lib, monkeypatch, D):
    """decim_shared_kernel (csdr's two decimations of the reference, 45 and 50: every input sample converted once per wave, the next
    output's prefix sum handed to the neighbour lane) and decim_systolic_kernel (rtl_fsk's in-process decimations 6 ... 30: an output's
    running sum travels through the lanes that hold its blocks) against decim_kernel (PIRIP_DECIM_SHARED=0): the same float32
    operations on the same operands in the same order, so f32 and s16 outputs are identical words -- random bytes, 5 streams, a
    length that leaves a partial tile."""
    import torch
    import pirip_amd
    rng = np.random.default_rng(D)
    B, n_in = 5, D * 2000 + 91
    host = rng.integers(0, 256, (B, n_in, 2), dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for tag, env in (("shared", "1"), ("per_output", "0")):
        monkeypatch.setenv("PIRIP_DECIM_SHARED", env)
        for s16 in (False, True):
            dec = pirip_amd.HipDecim(D, 0.05, out_s16=s16)
            n_out = dec.nout(n_in)
            o = torch.zeros((B, n_out, 2), dtype=torch.int16 if s16 else torch.float32, device="cuda")
            dec.batch(dev.data_ptr(), n_in * 2, n_in, o.data_ptr(), n_out * (4 if s16 else 8), B, st)
            torch.cuda.synchronize()
            outs[(tag, s16)] = o.cpu().numpy()
    for s16 in (False, True):
        a, b = outs[("shared", s16)], outs[("per_output", s16)]
        assert np.array_equal(a.view(np.uint16 if s16 else np.uint32), b.view(np.uint16 if s16 else np.uint32)), ("s16" if s16 else "f32")
    assert np.abs(outs[("shared", False)]).max() > 0.01
    # random lengths (from a single output to several tiles), even byte offsets, ragged strides, 1 .. 4 streams: the two kernels again word for word
    for draw in range(8):
        n_out_want = int(rng.integers(1, 900)) if draw else 1
        n_in = (n_out_want - 1) * D + 80 + int(rng.integers(0, D))
        Bq, off, pad = int(rng.integers(1, 5)), 2 * int(rng.integers(0, 9)), 2 * int(rng.integers(0, 7))
        stride = 2 * n_in + pad
        raw = torch.from_numpy(rng.integers(0, 256, off + Bq * stride + 64, dtype=np.uint8)).cuda()
        got = {}
        for tag, env in (("shared", "1"), ("per_output", "0")):
            monkeypatch.setenv("PIRIP_DECIM_SHARED", env)
            dec = pirip_amd.HipDecim(D, 0.05, out_s16=False)
            n_out = dec.nout(n_in)
            assert n_out == n_out_want
            o = torch.full((Bq, n_out, 2), 3.0, dtype=torch.float32, device="cuda")
            dec.batch(raw.data_ptr() + off, stride, n_in, o.data_ptr(), n_out * 8, Bq, st)
            torch.cuda.synchronize()
            got[tag] = o.cpu().numpy()
        assert np.array_equal(got["shared"].view(np.uint32), got["per_output"].view(np.uint32)), (D, draw, n_in, Bq, off, pad)
