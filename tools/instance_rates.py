#!/usr/bin/env python3
"""tools/instance_rates.py -- throughput of every wave-kernel instance family (and of the general kernel on the same shape) at full
occupancy on one GPU: a table for DESIGN.md / profiles/. Synthetic FSK from the product's CPU modulator + mild noise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import pirip_amd
import bench_configs

A = pirip_amd
FMT = {A.IN_CU8_FSKDEMOD: "u8 -d", A.IN_CU8_CSDR: "u8 csdr", A.IN_CS16: "s16", A.IN_CF32: "f32"}
SHAPES = [  # Fs, Rs, M, P, format, mask spacing, f1, shift, streams
    (240000, 10000, 2, 24, A.IN_CU8_FSKDEMOD, 0, 5000, 10000, 6144),
    (240000, 10000, 2, 24, A.IN_CU8_CSDR, 0, 5000, 10000, 6144),
    (240000, 10000, 2, 8, A.IN_CU8_FSKDEMOD, 0, 5000, 10000, 6144),
    (240000, 10000, 2, 6, A.IN_CU8_CSDR, 0, 5000, 10000, 6144),
    (240000, 10000, 2, 6, A.IN_CU8_CSDR, 10000, 10000, 10000, 6144),
    (240000, 10000, 4, 8, A.IN_CU8_FSKDEMOD, 0, 10000, 10000, 4096),
    (240000, 10000, 4, 8, A.IN_CU8_FSKDEMOD, 10000, 10000, 10000, 4096),
    (240000, 10000, 4, 6, A.IN_CU8_CSDR, 10000, 10000, 10000, 4096),
    (40000, 1000, 2, 8, A.IN_CS16, 0, 1000, 2000, 4096),
    (40000, 1000, 2, 10, A.IN_CF32, 0, 1000, 2000, 3072),
    (40000, 1000, 4, 10, A.IN_CF32, 2000, 1000, 2000, 2048),
    (40000, 1000, 4, 8, A.IN_CS16, 2000, 1000, 2000, 4096),
    (200000, 10000, 2, 10, A.IN_CF32, 10000, 10000, 10000, 3072),
    (200000, 10000, 4, 10, A.IN_CF32, 10000, 10000, 10000, 3072),
    (180000, 10000, 4, 9, A.IN_CF32, 10000, 10000, 10000, 3072),
    (100000, 10000, 2, 10, A.IN_CF32, 0, 10000, 10000, 4096),        # Ndft = 128 (README.md:196)
    (100000, 10000, 4, 10, A.IN_CF32, 10000, 10000, 10000, 4096),
    (80000, 10000, 2, 8, A.IN_CF32, 0, 10000, 10000, 4096),          # Ndft = 128 (README.md:172)
    (80000, 10000, 4, 8, A.IN_CF32, 8000, 8000, 8000, 4096),
]
# shapes without a wave instance (one workgroup per stream on the general kernel), swept over threads per stream
GENERAL_ONLY = [
    (240000, 1000, 2, 15, A.IN_CU8_CSDR, 0, 11000, 2000, 1024),      # rtl_fsk -r 1000 at 240 kS/s: Ts = 240, Ndft = 4096 (README.md:152,184)
    (240000, 1000, 4, 15, A.IN_CU8_CSDR, 2000, 11000, 2000, 1024),   # ... -m 4 --mask 2000 (README.md:239)
]


def rate(shape, kernel):
    Fs, Rs, M, P, fmt, mask, f1, shift, B = shape
    if kernel == "general":
        os.environ["PIRIP_KERNEL"] = "general"
        B = max(256, B // 2) if Fs // Rs < 100 else B
    else:
        os.environ.pop("PIRIP_KERNEL", None)
    if kernel == "block":
        B = int(os.environ.get("PIRIP_RATES_BLOCK_STREAMS", "2048"))
    Ts = Fs // Rs
    nsamp = (200 if Ts < 100 else 24) * 50 * Ts
    L = A.lib()
    x, _ = bench_configs.modulate(L, Fs, Rs, M, f1, shift, nsamp // Ts + 50, 7)
    x = x[:nsamp] + 0.2 * np.random.default_rng(3).standard_normal((nsamp, 2)).astype(np.float32)
    if fmt in (A.IN_CU8_FSKDEMOD, A.IN_CU8_CSDR):
        host = np.clip(np.rint(127.0 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
    elif fmt == A.IN_CS16:
        host = np.clip(np.rint(8000.0 * x.astype(np.float64)), -32768, 32767).astype(np.int16)
    else:
        host = x.astype(np.float32)
    bps = host.itemsize * 2
    dev = torch.from_numpy(host).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
    h = A.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=min(Fs // 2 - 1000, 90000), mask=mask, in_format=fmt, nstreams=B)
    assert h.kernel() == kernel, (shape, h.kernel())
    maxf = h.max_frames_for(nsamp)
    nb = 50 * (1 if M == 2 else 2)
    bits = torch.zeros((B, maxf, nb), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream()
    run = lambda: h.demod_batch(dev.data_ptr(), nsamp * bps, nsamp, bits.data_ptr(), maxf * nb, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 3
    e0.record(st)
    for _ in range(it):
        run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return float(cons.sum()) / ms / 1e6, B


if __name__ == "__main__":
    print("# Fs      Rs     M  P   input     estimator  | wave kernel: streams  G samples/s | general kernel: streams  G samples/s")
    wave_only = os.environ.get("PIRIP_RATES_WAVE_ONLY") is not None      # (profiling passes: tools/profile.sh instances)
    flt = os.environ.get("PIRIP_RATES_FILTER")            # e.g. "240000:4:8,240000:4:6": only these Fs:M:P shapes
    keep = lambda sh: not flt or f"{sh[0]}:{sh[2]}:{sh[3]}" in flt.split(",")
    for sh in ([] if os.environ.get("PIRIP_RATES_GENERAL_ONLY") else [s_ for s_ in SHAPES if keep(s_)]):
        w, bw = rate(sh, "wave")
        g, bg = (0.0, 0) if wave_only else rate(sh, "general")
        print(f"{sh[0]:7d} {sh[1]:6d} {sh[2]:2d} {sh[3]:3d}   {FMT[sh[4]]:<8s}  {'mask %d' % sh[5] if sh[5] else 'peak':<10s} | {bw:6d} {w:12.1f} | {bg:6d} {g:12.1f}", flush=True)
    if not wave_only:
        print("# Ts = 240 / Ndft = 4096 (rtl_fsk -r 1000 at 240 kS/s): the block instance (fsk_demod_block.hip, 256 threads per stream), then the general kernel")
        print("# at 64 / 128 / 256 / 384 / 512 threads per stream (PIRIP_GENERAL_THREADS) and at its default: G samples/s")
        for sh in GENERAL_ONLY:
            b, nb = rate(sh + (), "block")
            print(f"{sh[0]:7d} {sh[1]:6d} {sh[2]:2d} {sh[3]:3d}   {FMT[sh[4]]:<8s}  {'mask %d' % sh[5] if sh[5] else 'peak':<10s} | block instance, {nb} streams: {b:8.2f}", flush=True)
            if os.environ.get("PIRIP_RATES_BLOCK_ONLY"):
                continue
            rs = []
            for nt in ("64", "128", "256", "384", "512", None):
                if nt:
                    os.environ["PIRIP_GENERAL_THREADS"] = nt
                else:
                    os.environ.pop("PIRIP_GENERAL_THREADS", None)
                rs.append(rate(sh + (), "general")[0])
            print(f"{sh[0]:7d} {sh[1]:6d} {sh[2]:2d} {sh[3]:3d}   {FMT[sh[4]]:<8s}  {'mask %d' % sh[5] if sh[5] else 'peak':<10s} | " + "  ".join(f"{r:8.2f}" for r in rs), flush=True)
