#!/usr/bin/env python3
"""tools/ab_rates.py LIB_A LIB_B [Fs:M:P ...] -- A/B comparison of two builds of libpirip_hip.so on wave instances: the two libraries
are loaded in separate processes alternately (A B A B ...), several stream counts each, many launches per measurement; clocks and
power state drift by a few per cent between runs on this box, so only interleaved repeats are comparable."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import torch
    import pirip_amd
    import bench_configs
    Fs, M, P = (int(v) for v in sys.argv[2].split(":"))
    Rs, fmt = 10000, pirip_amd.IN_CU8_FSKDEMOD
    Ts, nsamp = Fs // Rs, 240000
    x, _ = bench_configs.modulate(pirip_amd.lib(), Fs, Rs, M, 10000 if M == 4 else 5000, 10000, nsamp // Ts + 50, 7)
    x = x[:nsamp] + 0.2 * np.random.default_rng(3).standard_normal((nsamp, 2)).astype(np.float32)
    host = np.clip(np.rint(127.0 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
    out = {}
    for B in (int(b) for b in sys.argv[3].split(",")):
        dev = torch.from_numpy(host).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
        h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=60000 if M == 4 else 25000, in_format=fmt, nstreams=B)
        h.set_bit_packing(True)
        maxf = h.max_frames_for(nsamp)
        nby = (50 * (1 if M == 2 else 2) + 7) // 8
        bits = torch.zeros((B, maxf, nby), dtype=torch.uint8, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream()
        run = lambda: h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * nby, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 40
        e0.record(st)
        for _ in range(it):
            run()
        e1.record(st); torch.cuda.synchronize()
        out[B] = float(cons.sum()) / (e0.elapsed_time(e1) / it) / 1e6
        kn = h.kernel_name()
        del h, dev, bits
    print("ABRATE " + json.dumps({"kernel": kn, "rates": out}))


def main():
    libs = sys.argv[1:3]
    shapes = sys.argv[3:] or ["240000:4:8"]
    counts = os.environ.get("AB_STREAMS", "4096,6144,12288")
    for sh in shapes:
        print(f"## {sh} (Fs:M:P), streams {counts}: G samples/s, runs interleaved A B A B A B")
        for rep in range(3):
            for tag, lib in zip("AB", libs):
                env = dict(os.environ, PIRIP_HIP_LIB=os.path.abspath(lib))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sh, counts], env=env, capture_output=True, text=True)
                ln = [l for l in r.stdout.splitlines() if l.startswith("ABRATE ")]
                if not ln:
                    print(tag, "failed", r.stderr[-300:]); continue
                d = json.loads(ln[0][7:])
                print(f"{tag} {os.path.basename(os.path.dirname(lib)):10s} " + "  ".join(f"{b}: {v:7.1f}" for b, v in d["rates"].items()) + "   " + d["kernel"], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
