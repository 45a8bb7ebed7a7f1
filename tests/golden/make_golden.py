"""Generate tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS REPO'S CPU oracle (oracle/fsk_oracle.c), not by
the reference: pirip's hot path lives in un-vendored codec2/csdr sources that are absent from
/root/reference and the reference holds no golden vectors for the path
(SURVEY.md 8c). They pin the oracle against regressions and give the GPU tests a fixed target;
they are NOT evidence of parity with codec2 ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob  # noqa: E402
import sigutil  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name), **kw)
    print(name, {k: getattr(v, "shape", v) for k, v in kw.items()})


def main():
    # cfg1: 10 frames, noise-free, timing offset 7 samples
    c = sigutil.CFG1
    u8, bits = sigutil.make_u8_stream(ob, c, 600, offset=7)
    rx = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    r = rx.demod(u8, ob.IN_CU8_FSKDEMOD)
    save("cfg1_clean.npz", iq_u8=u8, tx_bits=bits, bits=r["bits"], rx_filt=r["rx_filt"], stats=r["stats"],
         test_frame=ob.get_test_bits(100))
    # cfg1 noisy Eb/N0 = 8 dB, seed 3
    u8, bits = sigutil.make_u8_stream(ob, c, 600, seed=3, ebno_db=8.0, amp=20.0)
    rx = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    r = rx.demod(u8, ob.IN_CU8_FSKDEMOD)
    save("cfg1_noisy8dB.npz", iq_u8=u8, tx_bits=bits, bits=r["bits"], rx_filt=r["rx_filt"], stats=r["stats"])
    # cfg4: 4-FSK, 6 frames
    c = sigutil.CFG4
    u8, bits = sigutil.make_u8_stream(ob, c, 800, offset=3)
    rx = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    r = rx.demod(u8, ob.IN_CU8_FSKDEMOD)
    save("cfg4_clean.npz", iq_u8=u8, tx_bits=bits, bits=r["bits"], rx_filt=r["rx_filt"], stats=r["stats"])
    # kiss_fft: one 256-point and one 512-point transform of seeded data
    rng = np.random.default_rng(11)
    import ctypes as C
    L = ob.lib()
    L.kiss_fft_oracle_alloc.restype = C.c_void_p
    L.kiss_fft_oracle_alloc.argtypes = [C.c_int, C.c_int]
    L.kiss_fft_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    out = {}
    for n in (256, 512):
        x = rng.standard_normal((n, 2)).astype(np.float32)
        y = np.zeros_like(x)
        cfg = L.kiss_fft_oracle_alloc(n, 0)
        L.kiss_fft_oracle(cfg, x.ctypes.data, y.ctypes.data)
        out[f"x{n}"] = x; out[f"y{n}"] = y
    save("kiss_fft.npz", **out)
    # csdr decimator: taps + 2 blocks of u8 -> decimated s16
    L2 = ob.lib()
    ntaps = L2.oracle_firdes_filter_len(0.05)
    taps = np.zeros(ntaps, dtype=np.float32)
    L2.oracle_firdes_lowpass_f_hamming(taps.ctypes.data, ntaps, 0.5 / 45)
    u8 = rng.integers(0, 256, (45 * 40 + 84, 2)).astype(np.uint8)
    f = np.zeros(u8.shape, dtype=np.float32)
    L2.oracle_convert_u8_f(u8.ctypes.data, f.ctypes.data, u8.size)
    tp = np.zeros(84, dtype=np.float32); tp[:ntaps] = taps
    y = np.zeros((64, 2), dtype=np.float32)
    n = L2.oracle_fir_decimate_cc(f.ctypes.data, y.ctypes.data, u8.shape[0], 45, tp.ctypes.data, 84)
    s16 = np.zeros((n, 2), dtype=np.int16)
    L2.oracle_convert_f_s16(y.ctypes.data, s16.ctypes.data, 2 * n)
    save("csdr_decim45.npz", taps=taps, iq_u8=u8, y_f32=y[:n], y_s16=s16)


if __name__ == "__main__":
    main()
