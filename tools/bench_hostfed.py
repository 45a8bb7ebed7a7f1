#!/usr/bin/env python3
"""Rates of the headline configuration that are NOT bench.py's `value` (SURVEY.md 8d asks for them, labelled):
  * stream-count sweep B in {256, 1024, 4096, 16384}, device-resident input;
  * host-fed: pinned host u8 IQ -> hipMemcpyAsync on a copy stream -> demodulate on the compute stream, double
    buffered (the PCIe-inclusive rate of the batch API);
  * CLI: the `fsk_demod -d -p 24` process fed through a pipe (the correctness boundary's rate).
Profiling aid; prints one JSON object."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "pirip_amd", "bin")


def main():
    import torch
    import pirip_amd
    import bench
    res = {}
    nsamp = 1_200_000
    base, _ = bench.synth_base_streams(nsamp)
    one = torch.from_numpy(np.ascontiguousarray(base[2][:nsamp])).cuda()
    st = torch.cuda.current_stream()

    # ---- stream-count sweep, device resident ---------------------------------------------------------------
    sweep = {}
    for B in (256, 1024, 4096, 16384):
        ns = nsamp if B <= 4096 else nsamp // 4
        dev = one[:ns].unsqueeze(0).expand(B, ns, 2).contiguous()
        h = pirip_amd.HipDemod(bench.FS, bench.RS, 2, P=24, est_min=500, est_max=25000, in_format=0, nstreams=B)
        maxf = h.max_frames_for(ns)
        bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        run = lambda: h.demod_batch(dev.data_ptr(), ns * 2, ns, bits.data_ptr(), maxf * 50, 0, 0, 0, 0,
                                    nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            run()
        e1.record(st); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        sweep[str(B)] = {"samples_per_stream": ns, "ms": ms, "Gsamples_per_s": float(cons.sum()) / ms / 1e6}
        del dev, bits, h
    res["device_resident_sweep"] = sweep

    # ---- host-fed, double buffered -------------------------------------------------------------------------
    B, ns = 1024, nsamp
    host = [torch.from_numpy(np.ascontiguousarray(np.broadcast_to(base[2][:ns], (B, ns, 2)))).pin_memory() for _ in range(2)]
    devb = [torch.empty((B, ns, 2), dtype=torch.uint8, device="cuda") for _ in range(2)]
    h = pirip_amd.HipDemod(bench.FS, bench.RS, 2, P=24, est_min=500, est_max=25000, in_format=0, nstreams=B)
    maxf = h.max_frames_for(ns)
    bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]; done = [torch.cuda.Event() for _ in range(2)]
    nchunks = 12

    def pipeline():
        for k in range(nchunks):
            b = k & 1
            with torch.cuda.stream(copy_s):
                if k >= 2:
                    copy_s.wait_event(done[b])                 # the kernel that read this buffer has finished
                devb[b].copy_(host[b], non_blocking=True)
                ready[b].record(copy_s)
            comp_s.wait_event(ready[b])
            h.demod_batch(devb[b].data_ptr(), ns * 2, ns, bits.data_ptr(), maxf * 50, 0, 0, 0, 0,
                          nfr.data_ptr(), cons.data_ptr(), maxf, comp_s.cuda_stream)
            done[b].record(comp_s)
        torch.cuda.synchronize()
    pipeline()
    t0 = time.perf_counter(); pipeline(); dt = time.perf_counter() - t0
    res["host_fed_double_buffered"] = {"streams": B, "chunks": nchunks, "GB_per_s_h2d": nchunks * B * ns * 2 / dt / 1e9,
                                       "Gsamples_per_s": nchunks * B * ns / dt / 1e9}

    # ---- CLI through a pipe --------------------------------------------------------------------------------
    raw = np.ascontiguousarray(np.tile(base[2][:nsamp], (12, 1))).tobytes()      # 14.4 M samples = the 600 000-bit vector's size
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(BIN, "fsk_demod"), "--fsk_lower", "500", "--fsk_upper", "25000", "-d", "-p", "24",
                        "2", "240000", "10000", "-", "-"], input=raw, capture_output=True)
    dt = time.perf_counter() - t0
    res["cli_pipe"] = {"samples": len(raw) // 2, "bits_out": len(p.stdout), "seconds_incl_process_start": dt,
                       "Msamples_per_s": len(raw) / 2 / dt / 1e6, "rc": p.returncode}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
