import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sigutil, pirip_amd
from oracle import binding as ob
for M in (2, 4):
    Fs, Rs, P, f1, shift = 240000, 1000, 15, 11000, 2000
    c = dict(Fs=Fs, Rs=Rs, M=M, P=P, f1=f1, shift=shift, est_min=Rs // 2, est_max=Fs // 2 - Rs)
    rng = np.random.default_rng(Fs + M)
    bits = rng.integers(0, 2, 6000 * (1 if M == 2 else 2)).astype(np.uint8)
    x = sigutil.mod_complex(ob, c, bits)[rng.integers(0, Fs // Rs):]
    u8 = ob.quantise_cu8(x, amp=20.0)
    o = ob.OracleFsk(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"])
    h = pirip_amd.HipDemod(Fs, Rs, M, P=P, est_min=c["est_min"], est_max=c["est_max"], in_format=0, nstreams=1)
    ro = o.demod(u8, ob.IN_CU8_FSKDEMOD); rh = h.demod_host(u8)
    fo, fh = ro["rx_filt"].reshape(-1, M, 50), rh["rx_filt"].reshape(-1, M, 50)
    peak = np.abs(fo).max()
    e = np.abs(fo - fh) / peak
    print("M", M, "frames", ro["nframes"], "max err", e.max(), "timing err", np.abs(ro["stats"][:, 4] - rh["stats"][:, 4]).max())
    print(" per-frame max err", np.round(e.max(axis=(1, 2))[:12] * 1e5, 1), "...", np.round(e.max(axis=(1, 2))[-4:] * 1e5, 1))
    fr, m, s = np.unravel_index(np.argmax(e), e.shape)
    print(" worst at frame", fr, "tone", m, "sym", s, "oracle", fo[fr, m, s], "gpu", fh[fr, m, s], "f_est", ro["stats"][fr, :4], rh["stats"][fr, :4])
    print(" err by symbol (frame %d, tone %d) x1e5:" % (fr, m), np.round(e[fr, m] * 1e5, 1))
    rel = np.abs(fo - fh) / np.maximum(np.abs(fo), 1e-9)
    big = fo > 0.5 * peak
    print(" relative err where signal present: max", rel[big].max(), "mean signed", ((fh - fo) / fo)[big].mean())
