#!/usr/bin/env python3
"""tools/power_loop.py -- run the headline demodulator batch back to back for SECONDS (default 20) and print the rate; used with
tools/power_trace.sh, which samples rocm-smi meanwhile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pirip_amd, bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
B, nsamp = 16384, 240000
base, _ = bench.synth_base_streams(nsamp)
dev = torch.from_numpy(base[2][:nsamp]).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
h = pirip_amd.HipDemod(240000, 10000, 2, P=24, est_min=500, est_max=25000, nstreams=B)
maxf = h.max_frames_for(nsamp)
bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
run = lambda: h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * 50, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, 0)
run(); torch.cuda.synchronize()
print("loop start", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(10): run()
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
print(f"{n} batches in {dt:.2f} s: {n * B * nsamp / dt / 1e9:.1f} G samples/s", flush=True)
