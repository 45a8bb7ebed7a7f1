#!/usr/bin/env python3
"""tools/phase_split.py -- per-phase cycle split of the wave-per-stream demodulator instances.

Needs a library built with -DPIRIP_WAVE_TIMING (kept apart from the product build):
    make -C pirip_amd/csrc -j8 LIBDIR=../lib_timing EXTRA=-DPIRIP_WAVE_TIMING ../lib_timing/libpirip_hip.so
    PIRIP_HIP_LIB=pirip_amd/lib_timing/libpirip_hip.so python tools/phase_split.py [instance ...]
Each instance runs a full-occupancy batch with per-frame stats requested; the wave of stream nstreams/2 leaves its s_memtime
sums (shader cycles) in that stream's first stats row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import pirip_amd
import bench_configs

NAMES = ["wait for staged frame (vmcnt)", "estimator FFTs", "peak pick", "correlator",
         "DMA issue + hist copy + window sums + timing", "atan2 + decisions + outputs", "-", "loop overhead"]

# name: (Fs, Rs, M, P, in_format, est_max, streams, samples per stream, f1, shift)
INST = {
    "headline_2fsk_p24_u8d": (240000, 10000, 2, 24, pirip_amd.IN_CU8_FSKDEMOD, 25000, 6144, 240000, 5000, 10000),
    "2fsk_p24_u8csdr": (240000, 10000, 2, 24, pirip_amd.IN_CU8_CSDR, 25000, 6144, 240000, 5000, 10000),
    "2fsk_p8_u8d": (240000, 10000, 2, 8, pirip_amd.IN_CU8_FSKDEMOD, 25000, 6144, 240000, 5000, 10000),
    "4fsk_p8_u8d": (240000, 10000, 4, 8, pirip_amd.IN_CU8_FSKDEMOD, 60000, 4096, 240000, 10000, 10000),
    "ts40_2fsk_p8_s16": (40000, 1000, 2, 8, pirip_amd.IN_CS16, 20000, 4096, 80000, 1000, 2000),
    "ts40_2fsk_p8_f32": (40000, 1000, 2, 8, pirip_amd.IN_CF32, 20000, 2048, 80000, 1000, 2000),
    "ts40_4fsk_p10_f32": (40000, 1000, 4, 10, pirip_amd.IN_CF32, 18000, 2048, 80000, 1000, 2000),
    "ts20_4fsk_p10_f32": (200000, 10000, 4, 10, pirip_amd.IN_CF32, 90000, 3072, 200000, 10000, 10000),
}


def run(name):
    Fs, Rs, M, P, fmt, est_max, B, nsamp, f1, shift = INST[name]
    L = pirip_amd.lib()
    x, _ = bench_configs.modulate(L, Fs, Rs, M, f1, shift, nsamp // (Fs // Rs) + 50, 7)
    x = x[:nsamp]
    rng = np.random.default_rng(3)
    x = x + 0.2 * rng.standard_normal(x.shape).astype(np.float32)
    if fmt in (pirip_amd.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_CSDR):
        host = np.clip(np.rint(127.0 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
    elif fmt == pirip_amd.IN_CS16:
        host = np.clip(np.rint(8000.0 * x.astype(np.float64)), -32768, 32767).astype(np.int16)
    else:
        host = x.astype(np.float32)
    bps = host.itemsize * 2
    dev = torch.from_numpy(host).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
    h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=est_max, in_format=fmt, nstreams=B)
    maxf = h.max_frames_for(nsamp)
    nb = 50 * (1 if M == 2 else 2)
    bits = torch.zeros((B, maxf, nb), dtype=torch.uint8, device="cuda")
    stats = torch.zeros((B, maxf, pirip_amd.STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    for _ in range(2):
        h.demod_batch(dev.data_ptr(), nsamp * bps, nsamp, bits.data_ptr(), maxf * nb, 0, 0, stats.data_ptr(), maxf * pirip_amd.STATS_PER_FRAME,
                      nfr.data_ptr(), cons.data_ptr(), maxf, 0)
    torch.cuda.synchronize()
    t = stats[B // 2, 0].cpu().numpy().astype(np.float64)
    tot = t.sum()
    nf = int(nfr[B // 2])
    N = 50 * (Fs // Rs)
    print(f"## {name}: stream {B // 2} of {B}, {nf} frames of {N} samples; s_memtime ticks per frame {tot / nf:.0f}"
          f" ({tot / nf / N:.2f} per sample)")
    for n, v in zip(NAMES, t):
        if v:
            print(f"{n:<48} {v / nf:>9.0f} ticks/frame {100 * v / tot:5.1f} %")
    del h, dev, bits, stats


if __name__ == "__main__":
    if "timing" not in pirip_amd.lib_path():
        print("# warning: PIRIP_HIP_LIB does not point at a -DPIRIP_WAVE_TIMING build; the split will read back as zeros", file=sys.stderr)
    for nm in (sys.argv[1:] or list(INST)):
        run(nm)
