#!/usr/bin/env python3
"""tools/ldpc_precision.py -- CPU only: which of the product's FSK_LDPC precision choices moves frames across the decoding edge?

Synthetic soft decisions (4-FSK, Rician / Rayleigh tone magnitudes at a given Eb/N0 per channel bit, 50 symbols per demodulator
call) of continuous coded frames are received by oracle/ldpc_independent.c mode 2 (codec2's mapping as recalled, float32 soft bits,
double sum-product with libm phi: the stand-in for the reference) and by the mirror oracle (= the product's arithmetic) with its
experiment knobs: binary16 soft bits on / off, table phi (32 bins per octave) / finer tables / interpolated / exact. Reported: frames
delivered by one receiver only. Checker code only; nothing in the product imports this.
  python tools/ldpc_precision.py [--frames 40000] [--ebno 3.5]"""
import argparse
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CODE = os.path.join(ROOT, "pirip_amd", "data", "standin_256_512_4.code")
VARIANTS = [("product (binary16, table 32/octave, phi0 range)", 0), ("float32 soft bits, table 32", 1), ("binary16, table 64", 64 << 8), ("binary16, table 128", 128 << 8),
            ("binary16, table 32 interpolated", 4), ("binary16, table 32 interpolated from binary16 (base, slope) pairs", 4 | 16), ("binary16, table 32, binary16 (base, slope per unit x): fma(slope, x, base)", 4 | 32), ("binary16, table 16 interpolated", (16 << 8) | 4), ("binary16, table 8 interpolated", (8 << 8) | 4),
            ("binary16, table 4 interpolated", (4 << 8) | 4), ("binary16, exact phi", 2), ("float32, exact phi", 3),
            ("float32, exact phi, no range limits (messages up to 1e3)", 9), ("binary16, exact phi, no range limits", 8)]
_G = None


def _worker(args):
    seed, nfr, ebno = args
    from oracle import binding as ob
    import ctypes as C
    code = ob.parse_code_file(CODE)
    M, nsym = 4, 50
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = np.frombuffer(subprocess.run([framer, "--code", CODE, "-m", "4", "--testframes", str(nfr), "--bursts", "1", "--seq", "--source", "0x1", "/dev/zero", "-"],
                                      capture_output=True, check=True).stdout, dtype=np.uint8)
    pre = 200
    bits = fb[pre:]
    ns = bits.size // 2
    sym = bits[0:2 * ns:2] * 2 + bits[1:2 * ns:2]
    ncalls = ns // nsym
    sym = sym[:ncalls * nsym].reshape(ncalls, nsym)
    rng = np.random.default_rng(seed)
    esn0 = 2.0 * 10 ** (ebno / 10.0)
    z = (rng.normal(size=(ncalls, M, nsym)) + 1j * rng.normal(size=(ncalls, M, nsym))) / np.sqrt(2)
    ci, si = np.meshgrid(np.arange(ncalls), np.arange(nsym), indexing="ij")
    z[ci, sym, si] += np.sqrt(esn0)
    filt = (np.abs(z) * 0.37).astype(np.float32).reshape(ncalls, M * nsym)
    expected = np.stack([np.packbits(bits[f * 544 + 32:f * 544 + 32 + 256]) for f in range(nfr)])

    def delivered(st, pl):
        d = np.zeros(nfr, dtype=bool)
        for c in np.nonzero(st & 4)[0]:
            seq = int(pl[c, 1]) - 1
            if 0 <= seq < nfr and np.array_equal(pl[c], expected[seq]):
                d[seq] = True
        return d[2:-2]

    L = ob.lib()
    L.oracle_ldpc_experiment.argtypes = [C.c_int]
    st, pl, _ = ob.IndepLdpc(code, M, mode=3).rx(filt)
    out = {"recalled": delivered(st, pl)}
    st, pl, _ = ob.IndepLdpc(code, M, mode=2).rx(filt)
    out["recalled mapping, unlimited phi (mode 2)"] = delivered(st, pl)
    for name, flags in VARIANTS:
        L.oracle_ldpc_experiment(flags)
        st, pl, _ = ob.OracleLdpc(code, M).rx(filt)
        out[name] = delivered(st, pl)
    L.oracle_ldpc_experiment(0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40000)
    ap.add_argument("--ebno", type=float, default=3.5)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    per = 100
    jobs = [(1000 + i, per, a.ebno) for i in range(a.frames // per)]
    with mp.get_context("fork").Pool(a.procs) as pool:
        reps = pool.map(_worker, jobs)
    ref = np.concatenate([r["recalled"] for r in reps])
    print(f"# tools/ldpc_precision.py: 4-FSK synthetic soft decisions at Eb/N0 {a.ebno} dB (channel bit), {ref.size} frames scored; reference receiver = "
          f"oracle/ldpc_independent.c mode 3 (recalled codec2 mapping and phi0 range, float32 soft bits, double sum-product): FER {1 - ref.mean():.4f}")
    print("# mirror-oracle variant | FER | frames delivered by the variant only / by the reference only")
    for name in ["recalled mapping, unlimited phi (mode 2)"] + [v[0] for v in VARIANTS]:
        d = np.concatenate([r[name] for r in reps])
        print(f"{name:58s} | {1 - d.mean():.4f} | {int((d & ~ref).sum())} / {int((~d & ref).sum())}")


if __name__ == "__main__":
    main()
