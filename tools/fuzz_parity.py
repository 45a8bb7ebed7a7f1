#!/usr/bin/env python3
"""tools/fuzz_parity.py [--minutes 10] [--seed0 1] [--out FILE] -- randomised device-vs-oracle comparison, time-bounded:
every draw is a configuration fsk_create_hbr accepts (Ts 8 .. 240, any legal P, 2-/4-FSK, peak or mask estimator), an input format
(u8 `-d`, u8 csdr, s16, f32), a tone plan with a random offset, a random start phase, optionally AWGN (Eb/N0 5 .. 14 dB) and a sample
clock offset (+-50 .. 300 ppm), and a random split of the recording into 1 .. 4 calls (the unconsumed tail carried, as a reader does).
The device (whatever kernel the library picks for the shape) must give the oracle's frame counts, tone estimates, nin sequence and bits;
soft magnitudes, timing and SNR within the tolerances of tests/test_gpu_parity.py::_compare, whose rule for noisy inputs (a differing bit
only where the oracle's own decision margin is a near-tie) applies. A nin sequence that parts from the oracle's is accepted only at a
timing estimate within 5e-5 symbols of a decision threshold (tools/scale_check.py's "split"); everything after it is not compared.
Prints one line per draw that is not an exact pass and a summary; exit code 1 if anything is unexplained. The oracle is the checker."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


# shapes the reference's command lines use: (Ts, P, formats) with a specialised wave / block instance -- half the draws
REF_SHAPES = [(24, 24, (0, 3)), (24, 8, (0, 3)), (24, 6, (0, 3)), (40, 8, (1, 2)), (40, 10, (1, 2)), (20, 10, (2,)), (18, 9, (2,)),
              (10, 10, (2,)), (8, 8, (2,)), (240, 15, (0, 3))]


def draw(seed):
    rng = np.random.default_rng(seed)
    while True:
        if rng.random() < 0.5:
            Ts, P, fmts = REF_SHAPES[int(rng.integers(0, len(REF_SHAPES)))]
            Rs = int(rng.choice([600, 1000, 1200, 2400, 10000]))
            M = int(rng.choice([2, 4]))
            Fs = Ts * Rs
            k = int(rng.integers(1, 3))
            f1 = Rs * int(rng.integers(1, 3))
            if f1 + (M - 1) * k * Rs + Rs >= Fs // 2:
                continue
            return dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1 + int(rng.integers(-Rs // 4, Rs // 4)), shift=k * Rs, mask=int(k * Rs) if rng.random() < 0.3 and P != 24 else 0,
                        fmt=int(rng.choice(fmts)), ebno=None if rng.random() < 0.5 else float(rng.uniform(5.0, 14.0)),
                        ppm=0.0 if rng.random() < 0.8 else float(rng.choice([-1, 1]) * rng.uniform(50, 300)),
                        nframes=int(rng.integers(30, 120)), pieces=int(rng.integers(1, 5)), seed=seed)
        Ts = int(rng.choice([8, 10, 12, 16, 18, 20, 24, 24, 24, 32, 36, 40, 40, 48, 60, 64, 80, 96, 100, 120, 240, 240]))
        divs = [p for p in range(4, Ts + 1) if Ts % p == 0 and p <= 48]
        P = int(rng.choice(divs))
        Rs = int(rng.choice([100, 600, 1000, 1200, 2400, 4800, 10000]))
        M = int(rng.choice([2, 4]))
        Fs = Ts * Rs
        k = int(rng.integers(1, 3))
        f1 = Rs * int(rng.integers(1, 3))
        if f1 + (M - 1) * k * Rs + Rs >= Fs // 2:
            continue
        return dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1 + int(rng.integers(-Rs // 4, Rs // 4)), shift=k * Rs, mask=int(k * Rs) if rng.random() < 0.3 else 0,
                    fmt=int(rng.integers(0, 4)), ebno=None if rng.random() < 0.5 else float(rng.uniform(5.0, 14.0)),
                    ppm=0.0 if rng.random() < 0.8 else float(rng.choice([-1, 1]) * rng.uniform(50, 300)),
                    nframes=int(rng.integers(30, 120)), pieces=int(rng.integers(1, 5)), seed=seed)


def run_one(cfg, ob, A, sigutil, cmp):
    rng = np.random.default_rng(cfg["seed"] + 1)
    Fs, Rs, M, P = cfg["Fs"], cfg["Rs"], cfg["M"], cfg["P"]
    Ts = Fs // Rs
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=cfg["f1"], shift=cfg["shift"], est_min=Rs // 2, est_max=min(Fs // 2 - Rs, cfg["f1"] + M * cfg["shift"] + 2 * Rs))
    bits = rng.integers(0, 2, cfg["nframes"] * 50 * (1 if M == 2 else 2)).astype(np.uint8)
    x = sigutil.mod_complex(ob, c, bits)[int(rng.integers(0, Ts)):]
    if cfg["ebno"] is not None:
        x = sigutil.add_awgn(x, cfg["ebno"], c, rng)
    if cfg["ppm"]:
        r = 1.0 + cfg["ppm"] * 1e-6
        t = np.arange(int(x.shape[0] / r) - 2) * r
        i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
        x = ((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, x.shape[0] - 1)]).astype(np.float32)
    fmt = cfg["fmt"]
    amp = 25.0 if cfg["ebno"] is None else 12.0
    if fmt == 0:
        buf = ob.quantise_cu8(x, amp=amp); fo, fh = ob.IN_CU8_FSKDEMOD, A.IN_CU8_FSKDEMOD
    elif fmt == 3:
        buf = ob.quantise_cu8(x, amp=amp); fo, fh = ob.IN_CU8_CSDR, A.IN_CU8_CSDR
    elif fmt == 1:
        buf = np.clip(np.rint(x * 6000.0), -32768, 32767).astype(np.int16); fo, fh = ob.IN_CS16, A.IN_CS16
    else:
        buf = np.ascontiguousarray(x, dtype=np.float32); fo, fh = ob.IN_CF32, A.IN_CF32
    mask = cfg["mask"]
    o = ob.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], tone_spacing=mask if mask else 100, mask=bool(mask))
    try:
        h = A.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=fh, nstreams=1)
    except A.PiripError as e:
        return "skipped", str(e), None
    kern = h.kernel()
    ro = o.demod(buf, fo)
    cuts = sorted(rng.integers(0, buf.shape[0], cfg["pieces"] - 1).tolist()) + [buf.shape[0]]
    parts, carry, last = [], buf[:0], 0
    for cpos in cuts:
        piece = np.concatenate([carry, buf[last:cpos]]); last = cpos
        r = h.demod_host(piece)
        parts.append(r); carry = piece[r["consumed"]:]
    rh = {"nframes": sum(p["nframes"] for p in parts), "consumed": buf.shape[0] - carry.shape[0],
          "bits": np.concatenate([p["bits"] for p in parts]), "rx_filt": np.concatenate([p["rx_filt"] for p in parts]),
          "stats": np.concatenate([p["stats"] for p in parts])}
    h.close()
    tol = cmp.RX_FILT_TOL * max(1.0, Ts * 50 / 2400.0)
    note = None
    n = min(rh["nframes"], ro["nframes"])
    nin_o, nin_h = ro["stats"][:n, 6], rh["stats"][:n, 6]
    if not np.array_equal(nin_o, nin_h):
        f0 = int(np.nonzero(nin_o != nin_h)[0][0])
        t_o, t_d = float(ro["stats"][f0, 4]), float(rh["stats"][f0, 4])
        dist = min(abs(abs(t_o) - 0.25), abs(abs(t_d) - 0.25), abs(abs(t_o) - 0.5), abs(abs(t_d) - 0.5))
        if dist >= 5e-5:
            return "FAIL", f"nin sequence parts at frame {f0} away from a threshold (timing {t_o:+.6f} / {t_d:+.6f})", kern
        note = f"nin split at a threshold tie, frame {f0} of {n} (distance {dist:.1e})"
        keep = f0 if min(abs(abs(t_o) - 0.5), abs(abs(t_d) - 0.5)) < 5e-5 else f0 + 1
        for d in (ro, rh):
            for key in ("bits", "rx_filt", "stats"):
                d[key] = d[key][:keep]
            d["nframes"] = keep
        rh["consumed"] = ro["consumed"]
        ro["stats"] = ro["stats"].copy(); ro["stats"][:, 6] = rh["stats"][:, 6]      # (the split frame's nin_next is the known difference)
    # noise-free draws: the only differing bit there may be is a recording's very first decision (it starts mid-symbol: a fraction of a
    # symbol reaches that decision and the tone magnitudes tie) -- and it still has to pass the near-tie margin rule
    first_only = False
    if cfg["ebno"] is None and rh["bits"].shape == ro["bits"].shape and not np.array_equal(rh["bits"], ro["bits"]):
        d = np.argwhere(rh["bits"] != ro["bits"])
        if not all(fr == 0 and b < (1 if M == 2 else 2) for fr, b in d):
            return "FAIL", f"{len(d)} bit differences on a noise-free input, first at frame {d[0][0]} bit {d[0][1]}", kern
        first_only = True
    try:
        nfl = cmp._compare(ro, rh, tol=tol, allow_near_tie_flips=cfg["ebno"] is not None or first_only, M=M)
    except AssertionError as e:
        return "FAIL", str(e)[:300], kern
    if note:
        return "split", note, kern
    if first_only:
        return "first", "the recording's first decision (a fraction of a symbol) is a near-tie and differs", kern
    return ("near-tie", f"{nfl} near-tie bit differences of {rh['bits'].size}", kern) if nfl else ("exact", "", kern)


def batch_one(cfg, ob, A, sigutil, cmp):
    """pirip_hip_demod_batch over 1 .. 9 streams of one configuration, each with its own bits / tone offset / start phase / noise, laid out
    at a random stride, an optional frame cap: every stream's rows, frame count and consumed count against the oracle's"""
    import torch
    rng = np.random.default_rng(cfg["seed"] + 7)
    Fs, Rs, M, P = cfg["Fs"], cfg["Rs"], cfg["M"], cfg["P"]
    Ts = Fs // Rs
    B = int(rng.integers(1, 10))
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=cfg["f1"], shift=cfg["shift"], est_min=Rs // 2, est_max=min(Fs // 2 - Rs, cfg["f1"] + M * cfg["shift"] + 2 * Rs))
    fmt = cfg["fmt"]
    amp = 25.0 if cfg["ebno"] is None else 12.0
    nbits = cfg["nframes"] * 50 * (1 if M == 2 else 2)
    bufs = []
    for s_ in range(B):
        cs = dict(c, f1=c["f1"] + int(rng.integers(-Rs // 8, Rs // 8)))
        x = sigutil.mod_complex(ob, cs, rng.integers(0, 2, nbits).astype(np.uint8))[int(rng.integers(0, Ts)):]
        if cfg["ebno"] is not None:
            x = sigutil.add_awgn(x, cfg["ebno"], c, rng)
        bufs.append(x)
    nsamp = min(b.shape[0] for b in bufs)
    if fmt in (0, 3):
        host = [ob.quantise_cu8(b[:nsamp], amp=amp) for b in bufs]; fo = ob.IN_CU8_FSKDEMOD if fmt == 0 else ob.IN_CU8_CSDR
        fh = A.IN_CU8_FSKDEMOD if fmt == 0 else A.IN_CU8_CSDR
    elif fmt == 1:
        host = [np.clip(np.rint(b[:nsamp] * 6000.0), -32768, 32767).astype(np.int16) for b in bufs]; fo, fh = ob.IN_CS16, A.IN_CS16
    else:
        host = [np.ascontiguousarray(b[:nsamp], dtype=np.float32) for b in bufs]; fo, fh = ob.IN_CF32, A.IN_CF32
    bps = host[0].itemsize * 2
    mask = cfg["mask"]
    try:
        h = A.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], mask=mask, in_format=fh, nstreams=B)
    except A.PiripError as e:
        return "skipped", str(e), None
    kern = h.kernel()
    pad = int(rng.integers(0, 5)) * bps                      # stride in bytes: sample-aligned, not a round number
    stride = nsamp * bps + pad
    flat = np.zeros(B * stride + 64, dtype=np.uint8)
    for s_ in range(B):
        flat[s_ * stride: s_ * stride + nsamp * bps] = host[s_].view(np.uint8).reshape(-1)
    dev = torch.from_numpy(flat).cuda()
    maxf_all = h.max_frames_for(nsamp)
    cap = maxf_all if rng.random() < 0.6 else int(rng.integers(1, max(2, cfg["nframes"] // 2)))
    nb = h.Nbits
    bits = torch.zeros((B, cap, nb), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, cap, M * 50), dtype=torch.float32, device="cuda")
    stats = torch.zeros((B, cap, A.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    h.demod_batch(dev.data_ptr(), stride, nsamp, bits.data_ptr(), cap * nb, filt.data_ptr(), cap * M * 50, stats.data_ptr(), cap * A.STATS_PER_FRAME,
                  nfr.data_ptr(), cons.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    bits, filt, stats, nfr, cons = bits.cpu().numpy(), filt.cpu().numpy(), stats.cpu().numpy(), nfr.cpu().numpy(), cons.cpu().numpy()
    h.close()
    tol = cmp.RX_FILT_TOL * max(1.0, Ts * 50 / 2400.0)
    worst = "exact"; notes = []
    for s_ in range(B):
        o = ob.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], tone_spacing=mask if mask else 100, mask=bool(mask))
        ro = o.demod(host[s_], fo)
        n = min(ro["nframes"], cap)
        if n < ro["nframes"]:                                 # the cap: the oracle's first n frames and the samples they consumed
            ro = {"nframes": n, "consumed": None, "bits": ro["bits"][:n], "rx_filt": ro["rx_filt"][:n], "stats": ro["stats"][:n]}
        k = int(nfr[s_])
        rh = {"nframes": k, "consumed": int(cons[s_]), "bits": bits[s_, :k], "rx_filt": filt[s_, :k], "stats": stats[s_, :k]}
        if ro["consumed"] is None:
            # consumed after n frames = N (the first frame's nin from the reset state) + the nin_next column of the frames before the last
            ro["consumed"] = int(Ts * 50 + ro["stats"][:n - 1, 6].sum())
        if k != n:
            return "FAIL", f"stream {s_} of {B}: {k} frames, oracle {n} (cap {cap})", kern
        nin_o, nin_h = ro["stats"][:, 6], rh["stats"][:, 6]
        if not np.array_equal(nin_o, nin_h):
            f0 = int(np.nonzero(nin_o != nin_h)[0][0])
            t_o, t_d = float(ro["stats"][f0, 4]), float(rh["stats"][f0, 4])
            dist = min(abs(abs(t_o) - 0.25), abs(abs(t_d) - 0.25), abs(abs(t_o) - 0.5), abs(abs(t_d) - 0.5))
            if dist >= 5e-5:
                return "FAIL", f"stream {s_}: nin sequence parts at frame {f0} away from a threshold", kern
            worst = "split"; notes.append(f"stream {s_}: nin split at a threshold tie, frame {f0}")
            continue
        first_only = False
        if not np.array_equal(rh["bits"], ro["bits"]):
            d = np.argwhere(rh["bits"] != ro["bits"])
            first_only = all(fr == 0 and b < (1 if M == 2 else 2) for fr, b in d)
            if cfg["ebno"] is None and not first_only:
                return "FAIL", f"stream {s_}: {len(d)} bit differences on a noise-free input", kern
        try:
            nfl = cmp._compare(ro, rh, tol=tol, allow_near_tie_flips=cfg["ebno"] is not None or first_only, M=M)
        except AssertionError as e:
            return "FAIL", f"stream {s_} of {B}: " + str(e)[:260], kern
        if nfl and worst == "exact":
            worst = "first" if first_only else "near-tie"; notes.append(f"stream {s_}: {nfl} differing bits")
    return worst, "; ".join(notes), kern


def ldpc_one(seed, ob, A, sigutil):
    """FSK_LDPC receive (LLRs, unique-word search and tracking, LDPC decode, CRC16 -> status / payload / info records) against the mirror
    oracle, bit for bit, on the GPU demodulator's soft decisions: random M, oversampling, burst pattern (whole and cut-off bursts),
    Eb/N0 2 .. 10 dB, and the calls handed over in random chunks"""
    import subprocess
    rng = np.random.default_rng(seed)
    M = int(rng.choice([2, 4]))
    P = int(rng.choice([6, 8, 24])) if M == 2 else int(rng.choice([6, 8]))
    code_path = A.STANDIN_CODE
    code = ob.parse_code_file(code_path)
    c = dict(Fs=240000, Rs=10000, M=M, P=P, f1=10000, shift=10000, est_min=500, est_max=25000 if M == 2 else 60000)
    nfr = int(rng.integers(1, 5))
    fr = subprocess.run([os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer"), "--code", code_path, "-m", str(M), "--testframes", str(nfr), "--bursts", "1",
                         "--seq", "--source", hex(int(rng.integers(1, 15))), "/dev/zero", "-"], capture_output=True, check=True).stdout
    bits = np.frombuffer(fr, dtype=np.uint8)
    bursts = [bits if rng.random() < 0.7 else bits[: int(rng.integers(200, len(bits)))] for _ in range(int(rng.integers(1, 4)))]
    ebno = float(rng.uniform(2.0, 10.0))
    ts = 24
    segs = [np.zeros((int(rng.integers(500, 4000)), 2), dtype=np.float32)]
    for b in bursts:
        segs.append(sigutil.mod_complex(ob, c, b)); segs.append(np.zeros((int(rng.integers(2000, 14000)), 2), dtype=np.float32))
    x = np.concatenate(segs)
    sigma = np.sqrt(4.0 * ts / np.log2(M) / (10 ** (ebno / 10.0)) / 2.0)
    u8 = ob.quantise_cu8(x + rng.normal(0.0, sigma, x.shape).astype(np.float32), amp=14.0)
    dem = A.HipDemod(c["Fs"], c["Rs"], M, P=P, est_min=500, est_max=c["est_max"], in_format=A.IN_CU8_CSDR, nstreams=1)
    filt = dem.demod_host(u8)["rx_filt"]; dem.close()
    ws, wp, wi = ob.OracleLdpc(code, M).rx(filt)
    h = A.HipLdpc(code_path, M)
    gs, gp, gi, pos = [], [], [], 0
    while pos < len(filt):
        n = int(rng.choice([1, 2, 3, 7, 30, 64, 200]))
        s_, p_, i_ = h.rx_host(filt[pos:pos + n]); gs.append(s_); gp.append(p_); gi.append(i_); pos += n
    h.close()
    gs, gp, gi = np.concatenate(gs), np.concatenate(gp), np.concatenate(gi)
    if not (np.array_equal(gs, ws) and np.array_equal(gp, wp) and np.array_equal(gi, wi)):
        return "FAIL", f"FSK_LDPC records differ: M {M} P {P} Eb/N0 {ebno:.2f} {len(filt)} calls; first differing call {int(np.nonzero((gs != ws) | (gi != wi).any(axis=1))[0][0]) if len(gs) == len(ws) else 'count'}"
    return "exact", f"{int(((ws & 4) != 0).sum())} frames"


def chain_one(seed, ob, A, sigutil):
    """pirip_hip_fsk_ldpc_rx_batch (IQ -> records in one call, the demodulator instance writing bit LLRs and hard-decision words itself where it
    can): 1 .. 6 streams with their own start phase and noise, records against what the oracle's receiver makes of the soft magnitudes the
    unfused demodulator hands out for the same samples"""
    import subprocess
    import torch
    rng = np.random.default_rng(seed)
    M = int(rng.choice([2, 4]))
    P = int(rng.choice([6, 8, 24])) if M == 2 else int(rng.choice([6, 8]))
    fmt_h, fo_amp = (A.IN_CU8_CSDR, 14.0) if rng.random() < 0.5 else (A.IN_CU8_FSKDEMOD, 14.0)
    code_path = A.STANDIN_CODE
    code = ob.parse_code_file(code_path)
    c = dict(Fs=240000, Rs=10000, M=M, P=P, f1=10000, shift=10000)
    est_max = 25000 if M == 2 else 60000
    fr = subprocess.run([os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer"), "--code", code_path, "-m", str(M), "--testframes", str(int(rng.integers(1, 4))),
                         "--bursts", "1", "--seq", "--source", hex(int(rng.integers(1, 15))), "/dev/zero", "-"], capture_output=True, check=True).stdout
    bits = np.frombuffer(fr, dtype=np.uint8)
    segs = [np.zeros((int(rng.integers(900, 3000)), 2), dtype=np.float32)]
    for _ in range(int(rng.integers(1, 4))):
        segs += [sigutil.mod_complex(ob, c, bits), np.zeros((int(rng.integers(2000, 9000)), 2), dtype=np.float32)]
    segs.append(np.zeros((14400, 2), dtype=np.float32))
    x = np.concatenate(segs)
    ebno = float(rng.uniform(3.0, 9.0))
    sigma = np.sqrt(4.0 * 24 / np.log2(M) / (10 ** (ebno / 10.0)) / 2.0)
    B = int(rng.integers(1, 7))
    offs = [int(rng.integers(0, 48)) for _ in range(B)]
    nsamp = x.shape[0] - 48
    host = np.stack([ob.quantise_cu8(x[o:o + nsamp] + rng.normal(0.0, sigma, (nsamp, 2)).astype(np.float32), amp=fo_amp) for o in offs])
    want = []
    for s_ in range(B):
        dem = A.HipDemod(240000, 10000, M, P=P, est_min=500, est_max=est_max, in_format=fmt_h, nstreams=1)
        r = dem.demod_host(host[s_]); dem.close()
        want.append(ob.OracleLdpc(code, M).rx(r["rx_filt"]))
    dem = A.HipDemod(240000, 10000, M, P=P, est_min=500, est_max=est_max, in_format=fmt_h, nstreams=B)
    L = A.HipLdpc(code_path, M, nstreams=B)
    d = torch.from_numpy(host).cuda()
    maxf = dem.max_frames_for(nsamp)
    st = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda"); pl = torch.zeros((B, maxf, 32), dtype=torch.uint8, device="cuda")
    inf = torch.zeros((B, maxf, A.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    L.chain_batch(dem, d.data_ptr(), nsamp * 2, nsamp, st.data_ptr(), pl.data_ptr(), inf.data_ptr(), nfr.data_ptr(), cons.data_ptr(), maxf)
    torch.cuda.synchronize()
    fused = L.last_path_fused()
    nf = nfr.cpu().numpy(); st, pl, inf = st.cpu().numpy(), pl.cpu().numpy(), inf.cpu().numpy()
    L.close(); dem.close()
    for s_ in range(B):
        ws, wp, wi = want[s_]
        v = int(nf[s_])
        if v != len(ws) or not (np.array_equal(st[s_, :v], ws) and np.array_equal(pl[s_, :v], wp) and np.array_equal(inf[s_, :v], wi)):
            return "FAIL", f"chain records differ: M {M} P {P} fmt {fmt_h} Eb/N0 {ebno:.2f} B {B} stream {s_}: {v} calls, oracle {len(ws)}; fused {fused}"
        if st[s_, v:].any() or not (inf[s_, v:] == -1).all():
            return "FAIL", f"chain wrote beyond the stream's valid calls: stream {s_}"
    return "exact", "fused" if fused else "unfused"


def capture_one(seed, ob, A, sigutil):
    """pirip_hip_demod_capture (one recording demodulated frame-parallel with bit-for-bit verification) against the same library's read loop:
    every output array, the frame / sample counts and the state left behind, bit for bit -- random shape, noise, clock offset, work-slot count,
    segment length and number of pieces"""
    import test_capture as tc
    rng = np.random.default_rng(seed)
    cfg, fmt, mask = [(sigutil.CFG1, "u8", 0), (sigutil.CFG4, "u8", 0), (sigutil.CFG4, "u8", 10000), (dict(sigutil.CFG3, P=8), "s16", 0),
                      (dict(sigutil.CFG3, P=8), "f32", 0), (dict(sigutil.CFG1, P=8), "u8", 0)][int(rng.integers(0, 6))]
    ts = cfg["Fs"] // cfg["Rs"]
    nframes = int(rng.integers(600, 5000)) if ts == 24 else int(rng.integers(300, 1200))
    nbits = nframes * 50 * (1 if cfg["M"] == 2 else 2)
    ebno = None if rng.random() < 0.4 else float(rng.uniform(4.0, 12.0))
    ppm = 0.0 if rng.random() < 0.6 else float(rng.choice([-1, 1]) * rng.uniform(20e-6, 300e-6))
    fmts = {"u8": A.IN_CU8_FSKDEMOD, "s16": A.IN_CS16, "f32": A.IN_CF32}
    buf = tc._signal(ob, cfg, nbits, seed=seed, ppm=ppm, ebno_db=ebno, offset=int(rng.integers(0, ts)), fmt=fmt)
    hs = tc._mk(A, cfg, fmts[fmt], 1, mask)
    seq = hs.demod_host(buf)
    os.environ["PIRIP_CAPTURE_SEG_FRAMES"] = str(int(rng.choice([16, 16, 32, 64])))
    hc = tc._mk(A, cfg, fmts[fmt], int(rng.choice([8, 24, 64, 200])), mask)
    try:
        cap, reps = tc._capture(A, hc, buf, int(rng.integers(1, 4)))
        tc._same(cap, seq, "capture")
        sc_s, sf_s = tc._state(hs); sc_c, sf_c = tc._state(hc)
        if not (np.array_equal(sf_s.view(np.uint32), sf_c.view(np.uint32)) and np.array_equal(sc_s.view(np.uint32), sc_c.view(np.uint32))):
            return "FAIL", "state after the capture differs from the read loop's"
    except AssertionError as e:
        return "FAIL", f"capture {cfg['Fs']}/{cfg['Rs']} M {cfg['M']} P {cfg['P']} {fmt} mask {mask} Eb/N0 {ebno} ppm {ppm * 1e6:.0f}: {str(e)[:200]}"
    finally:
        os.environ.pop("PIRIP_CAPTURE_SEG_FRAMES", None); hs.close(); hc.close()
    return "exact", f"{max(r['passes'] for r in reps)} passes"


def decim_one(seed, ob, A):
    """csdr convert_u8_f | fir_decimate_cc D tbw | convert_f_s16 on the device against the oracle's scalar loop, bit for bit: random
    decimation, transition bandwidth (tap count), stream count, length, byte alignment and stride"""
    import torch
    rng = np.random.default_rng(seed)
    L = ob.lib()
    D = int(rng.integers(2, 65))
    tbw = float(rng.choice([0.05, 0.05, 0.02, 0.1, 0.2, 0.013]))
    B = int(rng.integers(1, 6))
    ntaps = L.oracle_firdes_filter_len(tbw)
    npad = ntaps + 3 - ((ntaps + 3) % 4)
    n_in = int(rng.choice([rng.integers(0, 3 * npad), rng.integers(npad, 60000)]))
    byte_off = int(rng.integers(0, 32)); stride = 2 * n_in + int(rng.integers(0, 40))
    host = rng.integers(0, 256, byte_off + B * stride + 64, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    tp = np.zeros(npad, dtype=np.float32)
    L.oracle_firdes_lowpass_f_hamming(tp.ctypes.data, ntaps, 0.5 / D)
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for s16 in (True, False):
        dec = A.HipDecim(D, tbw, out_s16=s16)
        n_out = dec.nout(n_in)
        want = 0 if n_in < npad else (n_in - npad) // D + 1
        if n_out != want:
            return "FAIL", f"decimator D {D} tbw {tbw} n_in {n_in}: {n_out} outputs, oracle {want}"
        o = torch.full((B, max(n_out, 1), 2), 77, dtype=torch.int16 if s16 else torch.float32, device="cuda")
        dec.batch(dev.data_ptr() + byte_off, stride, n_in, o.data_ptr(), max(n_out, 1) * (4 if s16 else 8), B, st)
        torch.cuda.synchronize()
        outs[s16] = o.cpu().numpy(); dec.close()
    if n_out == 0:
        ok = (outs[True] == 77).all() and (outs[False] == 77).all()
        return ("exact", "") if ok else ("FAIL", f"decimator D {D} n_in {n_in}: output written although no output is due")
    for s in range(B):
        u8 = np.ascontiguousarray(host[byte_off + s * stride: byte_off + s * stride + 2 * n_in]).reshape(n_in, 2)
        f = np.zeros(u8.shape, dtype=np.float32)
        L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
        y = np.zeros((n_in // D + 2, 2), dtype=np.float32)
        k = L.oracle_fir_decimate_cc(f.ctypes.data, y.ctypes.data, n_in, D, tp.ctypes.data, npad)
        q = np.zeros((n_out, 2), dtype=np.int16)
        L.oracle_convert_f_s16(y.ctypes.data, q.ctypes.data, 2 * n_out)
        if k != n_out or not np.array_equal(outs[False][s, :n_out].view(np.uint32), y[:n_out].view(np.uint32)) or not np.array_equal(outs[True][s, :n_out], q):
            return "FAIL", f"decimator D {D} tbw {tbw} ({ntaps} taps) n_in {n_in} B {B} byte_off {byte_off} stride {stride}: stream {s} differs"
    return "exact", ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", action="store_true", help="fuzz pirip_hip_demod_batch over several streams (strides, frame cap) instead of the one-stream host call")
    ap.add_argument("--capture", action="store_true", help="fuzz pirip_hip_demod_capture (frame-parallel, verified) against the read loop of the same library")
    ap.add_argument("--chain", action="store_true", help="fuzz pirip_hip_fsk_ldpc_rx_batch (IQ -> FSK_LDPC records, several streams) against demodulator + oracle receiver")
    ap.add_argument("--ldpc", action="store_true", help="fuzz the FSK_LDPC receiver (records bit-exact against the mirror oracle) instead of the demodulator")
    ap.add_argument("--decimator", action="store_true", help="fuzz the csdr front end (u8 -> decimated f32 / s16, bit-exact) instead of the demodulator")
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--seed0", type=int, default=1)
    ap.add_argument("--max-draws", type=int, default=1 << 30)
    a = ap.parse_args()
    import pirip_amd as A
    from oracle import binding as ob
    import sigutil
    import test_gpu_parity as cmp
    t0 = time.time()
    counts, kernels, frames = {}, {}, 0
    seed = a.seed0
    print(f"# tools/fuzz_parity.py --minutes {a.minutes} --seed0 {a.seed0}: one line per draw that is not an exact pass")
    while time.time() - t0 < a.minutes * 60 and seed - a.seed0 < a.max_draws:
        cfg = draw(seed)
        try:
            if a.decimator:
                res, msg = decim_one(seed, ob, A); kern = "decim"
            elif a.capture:
                res, msg = capture_one(seed, ob, A, sigutil); kern = "capture"; msg = "" if res == "exact" else msg
            elif a.chain:
                res, msg = chain_one(seed, ob, A, sigutil); kern = "chain " + msg if res == "exact" else "chain"; msg = "" if res == "exact" else msg
            elif a.ldpc:
                res, msg = ldpc_one(seed, ob, A, sigutil); kern = "ldpc"; msg = "" if res == "exact" else msg
            elif a.batch:
                res, msg, kern = batch_one(cfg, ob, A, sigutil, cmp)
            else:
                res, msg, kern = run_one(cfg, ob, A, sigutil, cmp)
        except Exception as e:                                  # a crash of the harness is a finding too
            res, msg, kern = "FAIL", f"{type(e).__name__}: {e}"[:300], None
        counts[res] = counts.get(res, 0) + 1
        if kern:
            kernels[kern] = kernels.get(kern, 0) + 1
        if res != "exact":
            print(f"{res:8s} seed {seed} {kern} Fs {cfg['Fs']} Rs {cfg['Rs']} M {cfg['M']} P {cfg['P']} mask {cfg['mask']} fmt {cfg['fmt']} "
                  f"Eb/N0 {cfg['ebno']} ppm {cfg['ppm']:.0f} pieces {cfg['pieces']}: {msg}", flush=True)
        seed += 1
    print(f"# {seed - a.seed0} draws in {time.time() - t0:.0f} s: " + ", ".join(f"{k} {v}" for k, v in sorted(counts.items())) +
          " | by kernel: " + ", ".join(f"{k} {v}" for k, v in sorted(kernels.items())))
    return 1 if counts.get("FAIL") else 0


if __name__ == "__main__":
    sys.exit(main())
