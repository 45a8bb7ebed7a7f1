#!/usr/bin/env python3
"""Measure the csdr front-end stage (convert_u8_f | fir_decimate_cc 45 | convert_f_s16, BASELINE
config 3's 1.8 MS/s side) on one GPU: input samples/s and achieved HBM GB/s against the 8 TB/s
roofline. Not the driver's bench (that is bench.py / config 2); a profiling aid for DESIGN.md.
  python tools/bench_decim.py [--streams 64] [--samples 90000000] [--iters 10] [--D 45]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--samples", type=int, default=45_000_000)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100, help="untimed launches first: the core clock needs a few hundred ms under load to ramp from idle")
    ap.add_argument("--D", type=int, default=45)
    args = ap.parse_args()
    import torch
    import pirip_amd
    dec = pirip_amd.HipDecim(args.D, 0.05, out_s16=True)
    n_in, B = args.samples, args.streams
    n_out = dec.nout(n_in)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randint(0, 256, (B, n_in, 2), dtype=torch.uint8, device="cuda", generator=g)
    y = torch.zeros((B, n_out, 2), dtype=torch.int16, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(args.warmup):
        dec.batch(x.data_ptr(), n_in * 2, n_in, y.data_ptr(), n_out * 4, B, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.iters):
        dec.batch(x.data_ptr(), n_in * 2, n_in, y.data_ptr(), n_out * 4, B, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    algo = B * (2.0 * n_in + 4.0 * n_out)
    out = {"stage": f"convert_u8_f | fir_decimate_cc {args.D} | convert_f_s16", "streams": B, "samples_per_stream": n_in,
           "kernel_ms": ms, "input_Msamples_per_s": B * n_in / ms / 1e3, "achieved_GBps": algo / ms / 1e6,
           "hbm_peak_GBps": 8000.0, "frac": algo / ms / 1e6 / 8000.0}
    # spot check against the oracle on the head of stream 0
    try:
        from oracle import binding as ob
        L = ob.lib()
        nchk = 45 * 2000 + 80
        u8 = x[0, :nchk].cpu().numpy()
        f = np.zeros(u8.shape, dtype=np.float32)
        L.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
        ntaps = L.oracle_firdes_filter_len(0.05)
        tp = np.zeros(80, dtype=np.float32)
        L.oracle_firdes_lowpass_f_hamming(tp.ctypes.data, ntaps, 0.5 / args.D)
        yy = np.zeros((nchk // args.D + 1, 2), dtype=np.float32)
        no = L.oracle_fir_decimate_cc(f.ctypes.data, yy.ctypes.data, nchk, args.D, tp.ctypes.data, 80)
        s16 = np.zeros((no, 2), dtype=np.int16)
        L.oracle_convert_f_s16(yy.ctypes.data, s16.ctypes.data, 2 * no)
        out["mismatches_vs_oracle"] = int((y[0, :no].cpu().numpy() != s16).sum())
    except Exception as e:
        out["mismatches_vs_oracle"] = f"unavailable: {e!r}"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
