#!/usr/bin/env python3
"""tools/phase_valu.py INSTANCE -- one full-occupancy launch of a wave instance in the bench's output mode (packed bits, no stats, no
soft magnitudes), for tools/phase_valu.sh to count instructions over. With a -DPIRIP_WAVE_TIMING library and PIRIP_WAVE_STOP=k
every frame ends after phase k (fsk_demod_wave.hip: PIRIP_T_MARK). Prints the frame count the counters have to be divided by."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import pirip_amd
import bench_configs
from phase_split import INST


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "headline_2fsk_p24_u8d"
    Fs, Rs, M, P, fmt, est_max, B, nsamp, f1, shift = INST[name]
    L = pirip_amd.lib()
    x, _ = bench_configs.modulate(L, Fs, Rs, M, f1, shift, nsamp // (Fs // Rs) + 50, 7)
    x = x[:nsamp] + 0.2 * np.random.default_rng(3).standard_normal((nsamp, 2)).astype(np.float32)
    if fmt in (pirip_amd.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_CSDR):
        host = np.clip(np.rint(127.0 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
    elif fmt == pirip_amd.IN_CS16:
        host = np.clip(np.rint(8000.0 * x.astype(np.float64)), -32768, 32767).astype(np.int16)
    else:
        host = x.astype(np.float32)
    bps = host.itemsize * 2
    dev = torch.from_numpy(host).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
    h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=500, est_max=est_max, in_format=fmt, nstreams=B)
    h.set_bit_packing(True)
    maxf = h.max_frames_for(nsamp)
    nby = (50 * (1 if M == 2 else 2) + 7) // 8
    bits = torch.zeros((B, maxf, nby), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    h.demod_batch(dev.data_ptr(), nsamp * bps, nsamp, bits.data_ptr(), maxf * nby, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, 0)
    torch.cuda.synchronize()
    print(f"PHASE_VALU instance {name} kernel {h.kernel_name()} streams {B} frames_total {int(nfr.sum())} stop {os.environ.get('PIRIP_WAVE_STOP', 'none')}")


if __name__ == "__main__":
    main()
