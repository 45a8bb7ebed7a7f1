import sys, os, numpy as np, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import sigutil, pirip_amd
from oracle import binding as ob
c = dict(sigutil.CFG3, P=8)
rng = np.random.default_rng(48)
z = (rng.normal(size=(2015, 2)) * 3000).astype(np.int16)      # white noise: every bin distinct
class Head(C.Structure):
    _fields_ = [("ints", C.c_int * 12), ("tc", C.c_float), ("est", C.c_int * 3), ("hann", C.c_void_p), ("Sf", C.c_void_p)]
os.environ["PIRIP_KERNEL"] = "wave"
o = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
h = pirip_amd.HipDemod(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"], in_format=2, nstreams=1)
ro = o.demod(z, 2); rh = h.demod_host(z)
Sfo = np.ctypeslib.as_array(C.cast(Head.from_address(o.h).Sf, C.POINTER(C.c_float)), shape=(512,)).copy()
hann = np.ctypeslib.as_array(C.cast(Head.from_address(o.h).hann, C.POINTER(C.c_float)), shape=(512,)).copy()
Sfh = h.get_Sf(0)
x = (z[:, 0].astype(np.float64) + 1j * z[:, 1]) / 750.0
mags = []
for j in range(6):
    X = np.fft.fftshift(np.fft.fft(x[256 * j:256 * j + 512] * hann))
    mags.append(np.abs(X))
mags = np.array(mags)
def smooth(order, w=None):
    S = np.zeros(512)
    for j in order: S = 0.9 * S + 0.1 * mags[j]
    return S
print("oracle vs numpy in-order", np.abs(smooth(range(6)) - Sfo).max() / Sfo.max())
print("device vs numpy in-order", np.abs(smooth(range(6)) - Sfh).max() / Sfo.max())
for order in ([1,0,3,2,5,4],):
    print("device vs numpy order", order, np.abs(smooth(order) - Sfh).max() / Sfo.max())
# per-bin: which single-FFT magnitude combos explain device? solve least squares weights per bin group
A = mags.T   # [512,6]
w, *_ = np.linalg.lstsq(A, Sfh, rcond=None)
print("device LSQ weights over the 6 FFTs (expect 0.059 0.066 0.073 0.081 0.09 0.1):", np.round(w, 4))
res = A @ w - Sfh
print("residual", np.abs(res).max() / Sfh.max())
# weights for even and odd bins separately, and by halves
for name, sel in (("bins%32<16", (np.arange(512) % 32) < 16), ("bins<256", np.arange(512) < 256), ("bins>=256", np.arange(512) >= 256)):
    w2, *_ = np.linalg.lstsq(A[sel], Sfh[sel], rcond=None); print(name, np.round(w2, 4))
