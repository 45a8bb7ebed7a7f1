import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import sigutil, pirip_amd
from oracle import binding as ob
c = dict(sigutil.CFG3, P=8)
rng = np.random.default_rng(48)
bits = rng.integers(0, 2, 6000).astype(np.uint8)
x = sigutil.mod_complex(ob, c, bits)[13:]
conv = lambda x: np.clip(np.trunc(x.astype(np.float64) * 8000.0), -32768, 32767).astype(np.int16)
y = sigutil.add_awgn(x, 9.0, c, rng)
nn = x.shape[0]
t = np.arange(int(nn / 1.0005) - 2) * 1.0005
i0 = np.floor(t).astype(int); fr = (t - i0)[:, None].astype(np.float32)
zc = conv(((1 - fr) * x[i0] + fr * x[np.minimum(i0 + 1, nn - 1)]).astype(np.float32))
for name, z in (("noisy", conv(y)), ("clock", zc)):
  for kern in ("wave", "general"):
    os.environ["PIRIP_KERNEL"] = kern
    o = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=2, nstreams=1)
    ro = o.demod(z, 2); rh = h.demod_host(z)
    print(name, kern, ro["nframes"], rh["nframes"], ro["consumed"], rh["consumed"])
    n = min(ro["nframes"], rh["nframes"])
    bad = np.where((ro["stats"][:n, 6] != rh["stats"][:n, 6]) | (ro["stats"][:n, 0] != rh["stats"][:n, 0])| (ro["stats"][:n, 1] != rh["stats"][:n, 1]))[0]
    print(" first mismatching frames:", bad[:10], " nin values seen:", np.unique(ro["stats"][:n, 6]))
    for f in list(bad[:2]):
        for g in (f - 1, f):
            print(g, "oracle", ro["stats"][g], "\n   hip  ", rh["stats"][g], " bitdiff", int((ro["bits"][g] != rh["bits"][g]).sum()),
              "filt err", float(np.abs(ro["rx_filt"][g] - rh["rx_filt"][g]).max() / np.abs(ro["rx_filt"][g]).max()))
