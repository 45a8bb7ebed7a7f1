"""pirip_hip_demod_capture: ONE long capture demodulated on many wavefronts (pirip_amd/csrc/capture.hip) must give what the
sequential read loop gives -- every output array bit for bit, the same frame and sample counts, the same state left behind --
whatever the signal does to the speculation (sample-clock slips, noise that moves the tone estimates, a different estimator,
a continuing stream). The sequential loop itself is pinned against the oracle in test_gpu_parity.py; here its first frames are
checked against the oracle once more so that the comparison cannot pass on two equally wrong results.

Reference: fsk_demod's argv names files (InputModemRawFile OutputOneBitPerByteFile, [UPSTREAM-RECALLED] codec2 fsk_demod.c usage; the
reference's own command lines give it pipes, /root/reference/README.md:105,109); VERDICT round 2 item 9."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import sigutil

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resample(x, ppm):
    """sample-clock offset by linear interpolation (as test_sample_clock_offset_exercises_nin_feedback does)"""
    n = x.shape[0]
    t = np.arange(int(n / (1 + abs(ppm)) - 2)) * (1 + ppm)
    i0 = np.floor(t).astype(int)
    fr = (t - i0)[:, None].astype(np.float32)
    return ((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, n - 1)]).astype(np.float32)


def _signal(ob, cfg, nbits, seed, ppm=0.0, ebno_db=None, offset=0, fmt="u8"):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    x = sigutil.mod_complex(ob, cfg, bits)
    if ppm:
        x = _resample(x, ppm)
    if ebno_db is not None:
        x = sigutil.add_awgn(x, ebno_db, cfg, rng)
    x = x[offset:]
    if fmt == "u8":
        return np.ascontiguousarray(ob.quantise_cu8(x, amp=20.0 if ebno_db is not None else 32.0))
    if fmt == "s16":
        return np.clip(np.trunc(x.astype(np.float64) * 5000.0), -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(x.astype(np.float32))


def _mk(pirip_amd, cfg, fmt, nstreams, mask=0):
    return pirip_amd.HipDemod(cfg["Fs"], cfg["Rs"], cfg["M"], P=cfg["P"], est_min=cfg["est_min"], est_max=cfg["est_max"], mask=mask,
                              in_format=fmt, nstreams=nstreams)


def _state(h):
    """everything pirip_hip_get_stream_state reports for stream slot 0 (struct pirip_stream_state: nin, norm_rx_timing, ppm, snr_est, SNRest,
    EbNodB, v_est, f_est[4], rx_sig_pow, rx_nse_pow -- 14 words) and its smoothed spectrum"""
    st = np.zeros(14, dtype=np.uint32)
    h.L.pirip_hip_get_stream_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert h.L.pirip_hip_get_stream_state(h.h, 0, st.ctypes.data) == 0
    return st.view(np.float32), h.get_Sf(0)


def _capture(pirip_amd, h, buf, pieces=1, maxf=None):
    """the capture through device pointers, optionally presented in pieces (unconsumed tail ahead of the next piece)"""
    import torch
    raw = np.ascontiguousarray(buf).reshape(buf.shape[0], -1)
    bps = raw.dtype.itemsize * raw.shape[1]
    n = raw.shape[0]
    maxf = maxf or h.max_frames_for(n)
    fb = h.Nbits
    bits = torch.zeros((maxf, fb), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((maxf, h.M * h.Nsym), dtype=torch.float32, device="cuda")
    stats = torch.zeros((maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    dev = torch.from_numpy(raw.view(np.uint8).reshape(n, bps)).cuda()
    cuts = [n * (i + 1) // pieces for i in range(pieces)]
    start, frames, reports = 0, 0, []
    for end in cuts:
        nf, cons, rep = h.demod_capture(dev.data_ptr() + start * bps, end - start, bits.data_ptr() + frames * fb,
                                        filt.data_ptr() + frames * h.M * h.Nsym * 4, stats.data_ptr() + frames * pirip_amd.STATS_PER_FRAME * 4,
                                        max_frames=maxf - frames)
        start += cons
        frames += nf
        reports.append(rep)
    torch.cuda.synchronize()
    return {"nframes": frames, "consumed": start, "bits": bits[:frames].cpu().numpy(), "rx_filt": filt[:frames].cpu().numpy(),
            "stats": stats[:frames].cpu().numpy()}, reports


def _same(a, b, what):
    assert a["nframes"] == b["nframes"] and a["consumed"] == b["consumed"], (what, a["nframes"], b["nframes"], a["consumed"], b["consumed"])
    assert np.array_equal(a["bits"], b["bits"]), what
    for k in ("rx_filt", "stats"):
        x, y = a[k].view(np.uint32), b[k].view(np.uint32)
        if not np.array_equal(x, y):
            bad = np.argwhere(x != y)
            raise AssertionError((what, k, bad[:5].tolist(), a[k][tuple(bad[0])], b[k][tuple(bad[0])]))


CASES = [
    # name, cfg, format, mask, bits, ppm, Eb/N0, slots, segment frames (PIRIP_CAPTURE_SEG_FRAMES), pieces
    ("headline clean", sigutil.CFG1, "u8", 0, 150000, 0.0, None, 64, 16, 1),
    ("headline noisy", sigutil.CFG1, "u8", 0, 150000, 0.0, 6.0, 64, 16, 1),
    ("headline very noisy (tone estimates move)", sigutil.CFG1, "u8", 0, 100000, 0.0, 1.0, 48, 16, 1),
    ("headline +50 ppm", sigutil.CFG1, "u8", 0, 150000, 50e-6, 9.0, 64, 16, 1),
    ("headline -120 ppm", sigutil.CFG1, "u8", 0, 150000, -120e-6, None, 64, 16, 1),
    ("headline +300 ppm", sigutil.CFG1, "u8", 0, 100000, 300e-6, 9.0, 32, 32, 1),
    ("headline in three pieces", sigutil.CFG1, "u8", 0, 150000, 20e-6, 8.0, 40, 16, 3),
    ("4-FSK", sigutil.CFG4, "u8", 0, 120000, 0.0, 9.0, 64, 16, 1),
    ("4-FSK mask estimator", sigutil.CFG4, "u8", 10000, 120000, 30e-6, 9.0, 64, 16, 1),
    ("Ts = 40 s16 (Ndft 512)", dict(sigutil.CFG3, P=8), "s16", 0, 20000, 0.0, 10.0, 24, 16, 1),
    ("Ts = 40 f32 +80 ppm", dict(sigutil.CFG3, P=8), "f32", 0, 20000, 80e-6, None, 24, 16, 1),
    # noisy enough for the timing loop to slip whole symbols: the passes after the first speculate on replicas a symbol apart
    ("headline 5 dB, long", sigutil.CFG1, "u8", 0, 600000, 0.0, 5.0, 256, 16, 1),
    ("4-FSK 4 dB", sigutil.CFG4, "u8", 0, 400000, 0.0, 4.0, 128, 16, 1),
    ("Ts = 40 s16 4 dB -40 ppm", dict(sigutil.CFG3, P=8), "s16", 0, 60000, -40e-6, 4.0, 96, 16, 1),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_capture_equals_the_sequential_read_loop(oracle, built_lib, case, monkeypatch):
    import pirip_amd
    name, cfg, fmt, mask, nbits, ppm, ebno, slots, segf, pieces = case
    fmts = {"u8": (pirip_amd.IN_CU8_FSKDEMOD, oracle.IN_CU8_FSKDEMOD), "s16": (pirip_amd.IN_CS16, oracle.IN_CS16),
            "f32": (pirip_amd.IN_CF32, oracle.IN_CF32)}
    buf = _signal(oracle, cfg, nbits, seed=len(name), ppm=ppm, ebno_db=ebno, offset=5, fmt=fmt)
    hs = _mk(pirip_amd, cfg, fmts[fmt][0], 1, mask)
    seq = hs.demod_host(buf)
    assert hs.kernel() == "wave"
    # the sequential loop against the oracle on the first frames (pinned in full elsewhere)
    nchk = min(buf.shape[0], 40 * hs.N)
    o = oracle.OracleFsk(cfg["Fs"], cfg["Rs"], cfg["M"], P=cfg["P"], est_min=cfg["est_min"], est_max=cfg["est_max"],
                         tone_spacing=mask if mask else 100, mask=bool(mask))
    ro = o.demod(buf[:nchk], fmts[fmt][1], want_filt=False)
    if ebno is None or ebno >= 6.0:     # (noisier: near-tie flips between the two float32 evaluation orders are possible, test_gpu_parity counts them)
        assert np.array_equal(seq["bits"][:ro["nframes"] - 1], ro["bits"][:ro["nframes"] - 1])

    monkeypatch.setenv("PIRIP_CAPTURE_SEG_FRAMES", str(segf))
    hc = _mk(pirip_amd, cfg, fmts[fmt][0], slots, mask)
    cap, reps = _capture(pirip_amd, hc, buf, pieces)
    _same(cap, seq, name)
    sc_s, sf_s = _state(hs)
    sc_c, sf_c = _state(hc)
    assert np.array_equal(sf_s.view(np.uint32), sf_c.view(np.uint32)), name
    assert np.array_equal(sc_s.view(np.uint32), sc_c.view(np.uint32)), (name, sc_s, sc_c)
    assert all(r["segments"] >= 3 for r in reps), reps          # the frame-parallel route ran
    if ppm == 0.0 and not mask and ebno is None and pieces == 1:
        assert reps[0]["passes"] == 1, reps                      # nothing to repair: every speculative start verified at once
    # the repair converges, it does not crawl: a few passes with little noise; where the timing loop slips whole symbols (below ~6 dB)
    # still far fewer passes than segments -- except at 1 dB, where it slips at random every few frames and every guess downstream of
    # a slip is void: exact all the same, but little faster than the read loop
    if ebno is None or ebno >= 6.0:
        assert all(r["passes"] <= 5 for r in reps), reps
    elif ebno >= 3.0:
        assert all(r["passes"] <= max(5, r["segments"] // 6) for r in reps), reps
    print(name, reps)

    # the sequential route of the same entry point (PIRIP_CAPTURE_SEQUENTIAL) and a general-kernel handle give the same again
    monkeypatch.setenv("PIRIP_CAPTURE_SEQUENTIAL", "1")
    hq = _mk(pirip_amd, cfg, fmts[fmt][0], 4, mask)
    capq, repq = _capture(pirip_amd, hq, buf, 1)
    _same(capq, seq, name + " (sequential route)")
    assert repq[0]["segments"] == 1


def test_capture_stops_at_the_output_capacity_like_the_read_loop(oracle, built_lib, monkeypatch):
    """max_frames is the room in the output arrays: the read loop stops there (pirip_hip_demod_batch's max_frames), the capture must
    stop at the same frame with the same state -- and continue from there on the next call."""
    import torch
    import pirip_amd
    cfg = sigutil.CFG1
    buf = _signal(oracle, cfg, 120000, seed=21, ppm=25e-6, ebno_db=9.0, offset=3)
    n = buf.shape[0]
    dev = torch.from_numpy(buf).cuda()
    monkeypatch.setenv("PIRIP_CAPTURE_SEG_FRAMES", "16")
    for limit in (1000, 777, 64):
        hs = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 1)
        bits = torch.zeros((limit, hs.Nbits), dtype=torch.uint8, device="cuda")
        stats = torch.zeros((limit, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
        nfr = torch.zeros(1, dtype=torch.int32, device="cuda")
        cons = torch.zeros(1, dtype=torch.int64, device="cuda")
        hs.demod_batch(dev.data_ptr(), 0, n, bits.data_ptr(), 0, 0, 0, stats.data_ptr(), 0, nfr.data_ptr(), cons.data_ptr(), limit, 0)
        torch.cuda.synchronize()
        assert int(nfr[0]) == limit
        hc = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 48)
        cap, reps = _capture(pirip_amd, hc, buf, 1, maxf=limit)
        assert cap["nframes"] == limit and cap["consumed"] == int(cons[0]), (limit, cap["nframes"], cap["consumed"], int(cons[0]))
        assert np.array_equal(cap["bits"], bits.cpu().numpy())
        assert np.array_equal(cap["stats"].view(np.uint32), stats.cpu().numpy().view(np.uint32))
        sc_s, sf_s = _state(hs)
        sc_c, sf_c = _state(hc)
        assert np.array_equal(sf_s.view(np.uint32), sf_c.view(np.uint32)) and np.array_equal(sc_s.view(np.uint32), sc_c.view(np.uint32)), limit
        # ... and both continue identically from where they stopped
        rest_s = hs.demod_host(buf[int(cons[0]):])
        rest_c, _ = _capture(pirip_amd, hc, buf[cap["consumed"]:], 1)
        _same(rest_c, rest_s, ("after the limit", limit))


def test_capture_with_frames_the_demodulator_skips(oracle, built_lib, monkeypatch):
    """Non-finite float samples make fsk_demod return early for the frames that contain them (the NaN guard): no timing
    update, zero bits. The capture's ppm recomputation has to skip exactly those rows, and the segments around them still verify."""
    import pirip_amd
    cfg = dict(sigutil.CFG3, P=8)
    buf = _signal(oracle, cfg, 30000, seed=5, ppm=60e-6, ebno_db=10.0, fmt="f32").copy()
    N = 2000
    for f in (7, 150, 151, 333):                                   # isolated frames and a pair, far from and near segment boundaries
        buf[f * N + 500, 0] = np.nan
    buf[420 * N + 3, 1] = np.inf
    hs = _mk(pirip_amd, cfg, pirip_amd.IN_CF32, 1)
    seq = hs.demod_host(buf)
    assert (seq["stats"][:, 9] == 0.0).sum() >= 4                  # those frames were skipped ...
    monkeypatch.setenv("PIRIP_CAPTURE_SEG_FRAMES", "16")
    hc = _mk(pirip_amd, cfg, pirip_amd.IN_CF32, 16)
    cap, reps = _capture(pirip_amd, hc, buf, 1)
    assert reps[0]["segments"] >= 3
    _same(cap, seq, "frames with non-finite samples")
    sc_s, sf_s = _state(hs)
    sc_c, sf_c = _state(hc)
    assert np.array_equal(sc_s.view(np.uint32), sc_c.view(np.uint32)), (sc_s, sc_c)


def test_capture_on_a_stream_of_the_callers(oracle, built_lib, monkeypatch):
    """hip_stream is honoured: all of the call's work (launches, copies, the host round trips' synchronisation) goes to the caller's stream"""
    import torch
    import pirip_amd
    cfg = sigutil.CFG1
    buf = _signal(oracle, cfg, 100000, seed=31, ppm=35e-6, ebno_db=7.0)
    hs = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 1)
    seq = hs.demod_host(buf)
    monkeypatch.setenv("PIRIP_CAPTURE_SEG_FRAMES", "16")
    hc = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 96)
    n = buf.shape[0]
    maxf = hc.max_frames_for(n)
    dev = torch.from_numpy(buf).cuda()
    bits = torch.zeros((maxf, hc.Nbits), dtype=torch.uint8, device="cuda")
    stats = torch.zeros((maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    nf, cons, rep = hc.demod_capture(dev.data_ptr(), n, bits.data_ptr(), 0, stats.data_ptr(), max_frames=maxf, stream=st.cuda_stream)
    st.synchronize()
    assert rep["segments"] >= 3 and nf == seq["nframes"] and cons == seq["consumed"]
    assert np.array_equal(bits[:nf].cpu().numpy(), seq["bits"])
    assert np.array_equal(stats[:nf].cpu().numpy().view(np.uint32), seq["stats"].view(np.uint32))


def test_capture_on_a_general_kernel_handle_takes_the_sequential_route(oracle, built_lib, monkeypatch):
    import pirip_amd
    cfg = dict(sigutil.CFG1, P=12)                                 # no wave instance for P = 12
    buf = _signal(oracle, cfg, 40000, seed=3, ebno_db=9.0)
    hs = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 1)
    assert hs.kernel() == "general"
    seq = hs.demod_host(buf)
    hc = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 16)
    cap, reps = _capture(pirip_amd, hc, buf, 1)
    _same(cap, seq, "general kernel")
    assert reps[0]["segments"] == 1 and reps[0]["passes"] == 1


def test_fsk_demod_on_a_file_uses_the_capture_route_and_matches_the_pipe(oracle, built_lib, tmp_path):
    """`fsk_demod ... file file` (the argv of README.md:105 with file names instead of `- -`) reads the whole file and demodulates it frame-parallel;
    the same bytes through a pipe take the read loop: identical output. The tool says which route it took with -v."""
    cfg = sigutil.CFG1
    buf = _signal(oracle, cfg, 120000, seed=11, ppm=40e-6, ebno_db=8.0)
    src = tmp_path / "in.u8"
    buf.tofile(src)
    exe = os.path.join(ROOT, "pirip_amd", "bin", "fsk_demod")
    argv = [exe, "-d", "-p", "24", "2", "240000", "10000"]
    p_file = subprocess.run(argv + [str(src), str(tmp_path / "out_file.bits")], capture_output=True, env=dict(os.environ, PIRIP_FSK_DEMOD_REPORT="1"))
    assert p_file.returncode == 0, p_file.stderr
    p_pipe = subprocess.run(argv + ["-", "-"], input=src.read_bytes(), capture_output=True)
    assert p_pipe.returncode == 0, p_pipe.stderr
    got = (tmp_path / "out_file.bits").read_bytes()
    assert got == p_pipe.stdout and len(got) > 100000
    assert b"capture:" in p_file.stderr and b"segments" in p_file.stderr, p_file.stderr
    # and with soft decisions (-s: floats) the same
    p_file = subprocess.run(argv[:1] + ["-s"] + argv[1:] + [str(src), str(tmp_path / "out_file.sd")], capture_output=True)
    p_pipe = subprocess.run(argv[:1] + ["-s"] + argv[1:] + ["-", "-"], input=src.read_bytes(), capture_output=True)
    assert p_file.returncode == 0 and (tmp_path / "out_file.sd").read_bytes() == p_pipe.stdout
    # complex s16 (-c, README.md:109's demodulator argv) and real s16 (fsk_demod's default input format: widened to complex on the way in)
    cfg3 = dict(sigutil.CFG3, P=8)
    s16 = _signal(oracle, cfg3, 40000, seed=12, ppm=-30e-6, ebno_db=10.0, fmt="s16")
    for name, data, args in (("cs16", s16, ["-c", "2", "40000", "1000"]), ("real s16", np.ascontiguousarray(s16[:, 0]), ["2", "40000", "1000"])):
        f = tmp_path / ("in_%s.raw" % name.replace(" ", "_"))
        data.tofile(f)
        out = tmp_path / "out.bits"
        p_file = subprocess.run([exe] + args + [str(f), str(out)], capture_output=True, env=dict(os.environ, PIRIP_FSK_DEMOD_REPORT="1"))
        p_pipe = subprocess.run([exe] + args + ["-", "-"], input=f.read_bytes(), capture_output=True)
        assert p_file.returncode == 0 and p_pipe.returncode == 0, (name, p_file.stderr, p_pipe.stderr)
        assert out.read_bytes() == p_pipe.stdout and len(p_pipe.stdout) > 30000, name
        assert b"capture:" in p_file.stderr, (name, p_file.stderr)
    # a file too short to be cut into segments takes the read loop inside the same call
    small = _signal(oracle, cfg, 6000, seed=13, ebno_db=9.0)
    fs = tmp_path / "small.u8"
    small.tofile(fs)
    p_file = subprocess.run(argv + [str(fs), str(tmp_path / "small.bits")], capture_output=True, env=dict(os.environ, PIRIP_FSK_DEMOD_REPORT="1"))
    p_pipe = subprocess.run(argv + ["-", "-"], input=fs.read_bytes(), capture_output=True)
    assert p_file.returncode == 0 and (tmp_path / "small.bits").read_bytes() == p_pipe.stdout and len(p_pipe.stdout) >= 5000
    assert b"1 segments of 0 frames" in p_file.stderr, p_file.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ebno,ppm", [(None, 0.0), (7.0, 0.0)])
def test_capture_with_the_band_only_estimator(oracle, built_lib, monkeypatch, ebno, ppm):
    """The frame-parallel route on a handle with the opt-in band-only estimator (pirip_hip_set_estimator_band_only): the segment chain
    is verified on the Sf bins that handle maintains; outputs, carried scalars and Sf inside the band equal the full estimator's
    sequential read loop word for word."""
    import pirip_amd
    cfg = sigutil.CFG1
    buf = _signal(oracle, cfg, 150000, seed=77, ppm=ppm, ebno_db=ebno, offset=9, fmt="u8")
    hs = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 1)
    seq = hs.demod_host(buf)
    monkeypatch.setenv("PIRIP_CAPTURE_SEG_FRAMES", "16")
    hc = _mk(pirip_amd, cfg, pirip_amd.IN_CU8_FSKDEMOD, 64)
    hc.set_estimator_band_only(True)
    cap, reps = _capture(pirip_amd, hc, buf, 1)
    _same(cap, seq, "band-only capture")
    sc_s, sf_s = _state(hs)
    sc_c, sf_c = _state(hc)
    assert np.array_equal(sf_s[128:160].view(np.uint32), sf_c[128:160].view(np.uint32))
    assert np.array_equal(sc_s.view(np.uint32), sc_c.view(np.uint32))
    assert all(r["segments"] >= 3 for r in reps) and all(r["passes"] <= 12 for r in reps), reps
