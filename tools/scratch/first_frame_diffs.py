"""Where do a fresh (reset-state) device pass and the oracle differ on the bench's noise-free streams? (frame / symbol index, margins)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import torch, pirip_amd
from oracle import binding as ob
nsamp = 120000
base, tx = bench.synth_base_streams(nsamp)
nd = 0
for plan in range(5):
    for off in range(24):
        buf = np.ascontiguousarray(base[plan][off:off + nsamp])
        h = pirip_amd.HipDemod(240000, 10000, 2, P=24, est_min=500, est_max=25000, nstreams=1)
        rh = h.demod_host(buf)
        o = ob.OracleFsk(240000, 10000, 2, P=24, est_min=500, est_max=25000)
        ro = o.demod(buf, ob.IN_CU8_FSKDEMOD)
        d = np.argwhere(rh["bits"] != ro["bits"])
        res = ob.put_test_bits(ro["bits"])
        if len(d) or res["errors"]:
            nd += len(d)
            for fr, b in d:
                f = ro["rx_filt"][fr]
                print("plan", plan, "off", off, "frame", fr, "sym", b, "oracle mags", f[b], f[50 + b], "gpu", rh["rx_filt"][fr][b], rh["rx_filt"][fr][50 + b], "timing", ro["stats"][fr, 4])
            print("plan", plan, "off", off, "oracle errors vs tx", res)
print("total diffs", nd)
