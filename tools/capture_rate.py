#!/usr/bin/env python3
"""tools/capture_rate.py -- one long capture: pirip_hip_demod_capture (frame-parallel, pirip_amd/csrc/capture.hip) against the
sequential read loop on the same samples (one stream = one wavefront). Prints one JSON line per case: both rates, how the
speculation went, and that every output row is identical.

  python tools/capture_rate.py [--samples 100000000] [--slots 2048]

Signals come from the device modulator (include/pirip_hip.h B2) with AWGN; the sample-clock-offset cases are resampled on the host
(shorter: numpy interpolation)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import pirip_amd as A
from pirip_amd.binding import synth_cu8


def synth(cfg, nsamp, ebno_db, seed):
    """nsamp + 7 samples modulated on the device in pieces of 40 modem frames (one modulator stream each, written back to back: the
    carrier phase restarts at every piece, which a non-coherent demodulator does not mind), then 7 samples dropped: a timing offset"""
    Fs, Rs, M, f1, shift = cfg["Fs"], cfg["Rs"], cfg["M"], cfg["f1"], cfg["shift"]
    Ts = Fs // Rs
    bps = 1 if M == 2 else 2
    piece_sym = 40 * 50
    piece = piece_sym * Ts
    npieces = (nsamp + 7 + piece - 1) // piece
    g = torch.Generator(device="cuda").manual_seed(seed)
    bits = torch.randint(0, 2, (npieces, piece_sym * bps), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty((npieces * piece, 2), dtype=torch.uint8, device="cuda")
    amp = 20.0
    sigma = 0.0
    if ebno_db is not None:
        # x = 2 exp(j phase): Ps = 4; Eb = Ps Ts / bps; N0 = Eb / ebno; sigma^2 = N0 / 2 per component (the modulator scales it by amp)
        n0 = 4.0 * Ts / bps / (10 ** (ebno_db / 10.0))
        sigma = float(np.sqrt(n0 / 2.0))
    synth_cu8(Fs, Rs, M, [f1] * npieces, shift, bits.data_ptr(), piece_sym * bps, piece_sym, out.data_ptr(), piece * 2, piece, amp=amp,
                sigma=sigma, seed=seed)
    torch.cuda.synchronize()
    return out[7:7 + nsamp].contiguous()


def resample_host(dev_u8, ppm):
    x = dev_u8.cpu().numpy().astype(np.float32) - 127.0
    n = x.shape[0]
    t = np.arange(int(n / (1 + abs(ppm)) - 2), dtype=np.float64) * (1 + ppm)
    i0 = np.floor(t).astype(np.int64)
    fr = (t - i0)[:, None].astype(np.float32)
    y = (1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, n - 1)]
    return torch.from_numpy(np.clip(np.rint(y + 127.0), 0, 255).astype(np.uint8)).cuda()


def run(name, cfg, dev, slots, mask=0):
    n = dev.shape[0]
    mk = lambda ns: A.HipDemod(cfg["Fs"], cfg["Rs"], cfg["M"], P=cfg["P"], est_min=cfg["est_min"], est_max=cfg["est_max"], mask=mask,
                               in_format=A.IN_CU8_FSKDEMOD, nstreams=ns)
    h1 = mk(1)
    maxf = h1.max_frames_for(n)
    outs = []
    for _ in range(2):
        outs.append((torch.zeros((maxf, h1.Nbits), dtype=torch.uint8, device="cuda"),
                     torch.zeros((maxf, A.STATS_PER_FRAME), dtype=torch.float32, device="cuda")))
    nfr = torch.zeros(1, dtype=torch.int32, device="cuda")
    cons = torch.zeros(1, dtype=torch.int64, device="cuda")
    # sequential: the batch entry point on a one-stream handle
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h1.demod_batch(dev.data_ptr(), 0, n, outs[0][0].data_ptr(), 0, 0, 0, outs[0][1].data_ptr(), 0, nfr.data_ptr(), cons.data_ptr(), maxf, 0)
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    nf_seq, cons_seq = int(nfr[0]), int(cons[0])
    # frame-parallel (second call of a fresh handle timed too: the first allocates the work area)
    times, rep = [], None
    for _ in range(2):
        hc = mk(slots)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nf, cn, rep = hc.demod_capture(dev.data_ptr(), n, outs[1][0].data_ptr(), 0, outs[1][1].data_ptr(), max_frames=maxf)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        # second timing on the same handle after a reset (work area in place)
        hc.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nf, cn, rep = hc.demod_capture(dev.data_ptr(), n, outs[1][0].data_ptr(), 0, outs[1][1].data_ptr(), max_frames=maxf)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t_cap = min(times)
    same = (nf == nf_seq and cn == cons_seq and bool(torch.equal(outs[0][0][:nf], outs[1][0][:nf]))
            and bool(torch.equal(outs[0][1][:nf].view(torch.int32), outs[1][1][:nf].view(torch.int32))))
    print(json.dumps({"case": name, "kernel": h1.kernel_name(), "samples": n, "frames": nf_seq,
                      "sequential_s": round(t_seq, 4), "sequential_Msamples_per_s": round(n / t_seq / 1e6, 1),
                      "capture_s": round(t_cap, 5), "capture_Msamples_per_s": round(n / t_cap / 1e6, 1), "speedup": round(t_seq / t_cap, 1),
                      "slots": slots, "report": rep, "identical_bits_and_stats_rows": same}), flush=True)
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100_000_000)
    ap.add_argument("--slots", type=int, default=2048)
    ap.add_argument("--only", default="", help="run only the cases whose name contains this")
    a = ap.parse_args()
    global run
    run_all = run
    run = lambda name, *rest, **kw: run_all(name, *rest, **kw) if a.only in name else True
    cfg1 = dict(Fs=240000, Rs=10000, M=2, P=24, f1=10000, shift=10000, est_min=500, est_max=25000)
    cfg4 = dict(Fs=240000, Rs=10000, M=4, P=8, f1=10000, shift=10000, est_min=500, est_max=60000)
    ok = True
    d = synth(cfg1, a.samples, None, 1)
    ok &= run("2-FSK -p 24, noise-free", cfg1, d, a.slots)
    d = synth(cfg1, a.samples, 9.0, 2)
    ok &= run("2-FSK -p 24, Eb/N0 9 dB", cfg1, d, a.slots)
    ok &= run("2-FSK -p 24, Eb/N0 9 dB, 256 slots", cfg1, d, 256)
    small = d[:min(a.samples, 24_000_000)]
    ok &= run("2-FSK -p 24, Eb/N0 9 dB, +30 ppm sample clock (24 M samples)", cfg1, resample_host(small, 30e-6), a.slots)
    ok &= run("2-FSK -p 24, Eb/N0 9 dB, -100 ppm sample clock (24 M samples)", cfg1, resample_host(small, -100e-6), a.slots)
    d = synth(cfg1, a.samples, 7.0, 5)
    ok &= run("2-FSK -p 24, Eb/N0 7 dB", cfg1, d, a.slots)
    d = synth(cfg1, a.samples, 6.0, 6)
    ok &= run("2-FSK -p 24, Eb/N0 6 dB", cfg1, d, a.slots)
    d = synth(cfg1, a.samples, 5.0, 3)
    ok &= run("2-FSK -p 24, Eb/N0 5 dB", cfg1, d, a.slots)
    d = synth(cfg1, a.samples, 4.0, 8)
    ok &= run("2-FSK -p 24, Eb/N0 4 dB", cfg1, d, a.slots)
    d = synth(cfg4, a.samples, 9.0, 4)
    ok &= run("4-FSK -p 8, Eb/N0 9 dB", cfg4, d, a.slots)
    ok &= run("4-FSK -p 8 --mask 10000, Eb/N0 9 dB", cfg4, d, a.slots, mask=10000)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
