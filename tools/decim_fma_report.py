#!/usr/bin/env python3
"""tools/decim_fma_report.py -- the decimator's ordering contract as a measurement (/root/reference/README.md:109,162).

The default tap loop (one multiply + one add per tap, ascending tap order) is the scalar csdr loop's float32 result bit for bit
(oracle/csdr_oracle.c). Upstream csdr is built -O3 -ffast-math [UPSTREAM-RECALLED], so the shipped binary's summation order is
its vectoriser's. This tool runs the two OPT-IN arithmetics (pirip_hip_decim_set_arith 1 = fused accumulate, 2 = affine map out
of the sum) against the default on (a) the committed fixture's input (tests/golden/csdr_decim45.npz), (b) 10^8 random bytes'
worth of samples, and reports rate and how many s16 outputs differ and by how much.
  python tools/decim_fma_report.py > profiles/r05_decim_fma_report.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import pirip_amd
    st = torch.cuda.current_stream()
    dec = pirip_amd.HipDecim(45, 0.05, out_s16=True)
    print("# decimator /45, 79 taps, s16 out: opt-in tap-loop arithmetics against the default (exact = scalar csdr loop bit for bit)")
    print("# input | mode | s16 outputs | differing | largest difference (LSB) | kernel ms | input G samples/s | fraction of 8 TB/s")
    cases = []
    fx = np.load(os.path.join(ROOT, "tests", "golden", "csdr_decim45.npz"))
    key = [k for k in fx.files if fx[k].dtype == np.uint8][0]
    u8 = np.ascontiguousarray(fx[key]).reshape(-1, 2)
    cases.append(("fixture csdr_decim45.npz (%d samples)" % u8.shape[0], torch.from_numpy(u8).cuda().unsqueeze(0), 1))
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    B, n = 64, 45_000_000 // 64 * 2                       # 64 x 1.4 M = 9 x 10^7 samples (1.8 x 10^8 random bytes)
    cases.append(("uniform random bytes, %d x %d samples" % (B, n), torch.randint(0, 256, (B, n, 2), dtype=torch.uint8, device="cuda", generator=g), B))
    # a modulated signal with noise (the decimator's real input): 2-FSK at 1.8 MS/s, amplitude 40 LSB + noise 20 LSB rms
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    sig = torch.stack([torch.cos(2 * np.pi * 1500.0 / 1.8e6 * t), torch.sin(2 * np.pi * 1500.0 / 1.8e6 * t)], dim=1)
    noisy = 127.5 + 40.0 * sig.unsqueeze(0) + 20.0 * torch.randn((B, n, 2), device="cuda", dtype=torch.float64, generator=g)
    cases.append(("tone + Gaussian noise, %d x %d samples" % (B, n), torch.clamp(torch.round(noisy), 0, 255).to(torch.uint8), B))
    del t, sig, noisy
    for name, x, nb in cases:
        n_in = x.shape[1]
        n_out = dec.nout(n_in)
        outs = {}
        for mode, mname in ((0, "exact"), (1, "fma"), (2, "fma_raw")):
            dec.set_arith(mode)
            y = torch.zeros((nb, n_out, 2), dtype=torch.int16, device="cuda")
            for _ in range(3):
                dec.batch(x.data_ptr(), n_in * 2, n_in, y.data_ptr(), n_out * 4, nb, st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 20 if nb > 1 else 5
            e0.record(st)
            for _ in range(it):
                dec.batch(x.data_ptr(), n_in * 2, n_in, y.data_ptr(), n_out * 4, nb, st.cuda_stream)
            e1.record(st); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            outs[mname] = y
            d = (y.int() - outs["exact"].int()).abs()
            rate = nb * n_in / ms / 1e6
            print(f"{name} | {mname:8s} | {y.numel()} | {int((d > 0).sum())} | {int(d.max())} | {ms:.3f} | {rate:.1f} | {rate * 1e9 * (2.0 + 4.0 / 45.0) / 1e9 / 8000.0:.3f}", flush=True)
    dec.set_arith(0)


if __name__ == "__main__":
    main()
