import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import sigutil, pirip_amd
from oracle import binding as ob
c = dict(sigutil.CFG3, P=8)
rng = np.random.default_rng(48)
bits = rng.integers(0, 2, 6000).astype(np.uint8)
x = sigutil.mod_complex(ob, c, bits)[13:]
conv = lambda x: np.clip(np.trunc(x.astype(np.float64) * 8000.0), -32768, 32767).astype(np.int16)
class Head(C.Structure):
    _fields_ = [("ints", C.c_int * 12), ("tc", C.c_float), ("est", C.c_int * 3), ("hann", C.c_void_p), ("Sf", C.c_void_p)]
for nfr in (1, 2, 5):
  z = conv(x)[:2000 * nfr + 15]
  for kern in ("wave", "general"):
    os.environ["PIRIP_KERNEL"] = kern
    o = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
    h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=2, nstreams=1)
    ro = o.demod(z, 2); rh = h.demod_host(z)
    Sfo = np.ctypeslib.as_array(C.cast(Head.from_address(o.h).Sf, C.POINTER(C.c_float)), shape=(512,)).copy()
    Sfh = h.get_Sf(0)
    d = np.where(Sfo != Sfh)[0]
    print(nfr, kern, "frames", ro["nframes"], rh["nframes"], "Sf mismatches", len(d), d[:20], "max rel", float(np.max(np.abs(Sfo - Sfh)) / np.max(Sfo)))
    if len(d): print("   e.g.", [(int(i), float(Sfo[i]), float(Sfh[i])) for i in d[:6]])
