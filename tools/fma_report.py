#!/usr/bin/env python3
"""tools/fma_report.py -- what the opt-in fused complex multiply in the estimator FFT (PIRIP_FFT_FMA=1, headline shape only)
changes: Sf bits, f_est / nin / decoded-bit sequences against the oracle (x86 no-FMA arithmetic) on clean, noisy and stress
inputs, next to the exact kernel. Run on the GPU box; the summary is kept as profiles/r02_fft_fma_report.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sigutil  # noqa: E402
import pirip_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402  (checker)

c = sigutil.CFG1
cases = [("clean 600k bits", dict(nbits=600000)),
         ("AWGN 12 dB", dict(nbits=100000, seed=1, ebno_db=12.0, random_bits=True, amp=18.0)),
         ("AWGN 8 dB", dict(nbits=100000, seed=2, ebno_db=8.0, random_bits=True, amp=18.0)),
         ("AWGN 5 dB", dict(nbits=100000, seed=3, ebno_db=5.0, random_bits=True, amp=18.0)),
         ("AWGN 3 dB", dict(nbits=100000, seed=4, ebno_db=3.0, random_bits=True, amp=14.0)),
         ("AWGN 1 dB", dict(nbits=100000, seed=5, ebno_db=1.0, random_bits=True, amp=10.0)),
         ("clipping, 12 dB", dict(nbits=60000, seed=6, ebno_db=12.0, random_bits=True, amp=70.0))]
print(f"{'case':<18} {'kernel':<6} {'frames':>7} {'f_est!=':>8} {'nin!=':>6} {'bits!=':>7} {'Sf bits differ':>15} {'max |dSf|/max Sf':>17}")
for name, kw in cases:
    u8, _ = sigutil.make_u8_stream(ob, c, **kw)
    o = ob.OracleFsk(c["Fs"], c["Rs"], 2, P=24, est_min=500, est_max=25000)
    ro = o.demod(u8, ob.IN_CU8_FSKDEMOD, want_filt=False)
    import ctypes as C

    o.l.oracle_fsk_get_Sf.restype = C.c_void_p; o.l.oracle_fsk_get_Sf.argtypes = [C.c_void_p]
    Sfo = np.ctypeslib.as_array(C.cast(o.l.oracle_fsk_get_Sf(o.h), C.POINTER(C.c_float)), shape=(256,)).copy()
    for kern, env in (("exact", "0"), ("fma", "1")):
        os.environ["PIRIP_FFT_FMA"] = env
        h = pirip_amd.HipDemod(c["Fs"], c["Rs"], 2, P=24, est_min=500, est_max=25000, nstreams=1)
        rh = h.demod_host(u8, want_filt=False)
        n = min(ro["nframes"], rh["nframes"])
        Sfh = h.get_Sf(0)
        print(f"{name:<18} {kern:<6} {rh['nframes']:>7} {int((ro['stats'][:n, :2] != rh['stats'][:n, :2]).any(axis=1).sum()):>8} "
              f"{int((ro['stats'][:n, 6] != rh['stats'][:n, 6]).sum()):>6} {int((ro['bits'][:n] != rh['bits'][:n]).sum()):>7} "
              f"{int((Sfo != Sfh).sum()):>15} {float(np.abs(Sfo - Sfh).max() / Sfo.max()):>17.2e}")
