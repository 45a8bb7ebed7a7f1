#!/usr/bin/env python3
"""tools/band_soak.py [--minutes 5] [--seed0 1] -- soak of the opt-in band-only estimator (pirip_hip_set_estimator_band_only) against the full
estimator of the same library: random tone plans inside the search range (edges included), noise or none, start offsets, amplitudes, input
format, the recording handed over in 1..5 uneven pieces, search range narrowed at random inside the band. Every output word (bits, soft
magnitudes, stats rows, frame and sample counts) and Sf inside the band must be identical; anything else is printed. GPU box only."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed0", type=int, default=1)
    a = ap.parse_args()
    import pirip_amd
    import sigutil
    from oracle import binding as ob            # signal generation only (the modulator and the u8 quantiser)
    print(f"# tools/band_soak.py --minutes {a.minutes} --seed0 {a.seed0}: one line per draw whose band-only outputs differ from the full estimator's")
    t0, n, bad, frames = time.time(), 0, 0, 0
    handles = {}
    while time.time() - t0 < a.minutes * 60:
        seed = a.seed0 + n
        rng = np.random.default_rng(seed)
        M = 2 if rng.random() < 0.6 else 4
        P, band_hz = (24, 30000) if M == 2 else (8, 60000)
        fmt = int(rng.integers(0, 2))
        lo = int(rng.choice([0, 300, 500, 1500]))
        hi = int(rng.choice([band_hz - 1, band_hz - 2000, 25000 if M == 2 else 55000]))
        shift = int(rng.choice([5000, 8000, 10000, 12000])) if M == 2 else int(rng.choice([9000, 10000, 12000, 14000]))
        span = shift * (M - 1)
        f1 = int(rng.integers(max(lo, 200) + 100, max(hi - span - 100, max(lo, 200) + 200)))
        ebno = None if rng.random() < 0.3 else float(rng.uniform(3.0, 12.0))
        cfg = dict(Fs=240000, Rs=10000, M=M, P=P, f1=f1, shift=shift)
        u8, _ = sigutil.make_u8_stream(ob, cfg, int(rng.integers(1500, 12000)) * (1 if M == 2 else 2), seed=seed, offset=int(rng.integers(0, 48)),
                                       ebno_db=ebno, random_bits=True, amp=float(rng.choice([8.0, 20.0, 32.0, 50.0])))
        key = (M, fmt, lo, hi)
        if key not in handles:
            f = pirip_amd.HipDemod(240000, 10000, M, P=P, est_min=lo, est_max=hi, in_format=[pirip_amd.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_CSDR][fmt], nstreams=1)
            b = pirip_amd.HipDemod(240000, 10000, M, P=P, est_min=lo, est_max=hi, in_format=[pirip_amd.IN_CU8_FSKDEMOD, pirip_amd.IN_CU8_CSDR][fmt], nstreams=1)
            b.set_estimator_band_only(True)
            assert "band-only" in b.kernel_name()
            handles[key] = (f, b)
        f, b = handles[key]
        f.reset(); b.reset()
        cuts = sorted(set(int(v) for v in rng.integers(1, u8.shape[0], int(rng.integers(0, 5))))) + [u8.shape[0]]
        pos, why = 0, None
        for c in cuts:
            if c <= pos:
                continue
            ra, rb = f.demod_host(u8[pos:c]), b.demod_host(u8[pos:c])
            if ra["nframes"] != rb["nframes"] or ra["consumed"] != rb["consumed"]:
                why = "frame / sample counts"
            elif not np.array_equal(ra["bits"], rb["bits"]):
                why = "bits"
            elif not np.array_equal(ra["rx_filt"].view(np.uint32), rb["rx_filt"].view(np.uint32)):
                why = "soft magnitudes"
            elif not np.array_equal(ra["stats"].view(np.uint32), rb["stats"].view(np.uint32)):
                why = "stats"
            if why:
                break
            frames += ra["nframes"]
            pos += ra["consumed"]
        if not why:
            w = 32 if M == 2 else 64
            if not np.array_equal(f.get_Sf(0)[128:128 + w].view(np.uint32), b.get_Sf(0)[128:128 + w].view(np.uint32)):
                why = "Sf inside the band"
        if why:
            bad += 1
            print(f"DIFF seed {seed} M {M} fmt {fmt} range {lo}..{hi} f1 {f1} shift {shift} Eb/N0 {ebno} pieces {len(cuts)}: {why}", flush=True)
        n += 1
    print(f"# {n} draws, {frames} frames in {time.time() - t0:.0f} s: {n - bad} identical, {bad} different")


if __name__ == "__main__":
    main()
