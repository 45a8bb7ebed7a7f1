#!/usr/bin/env python3
"""tools/update_hbm_traffic.py profiles/r04_b_headline_pmc.txt -- rewrites profiles/hbm_traffic.json (what bench.py quotes as
roofline.traffic and valu.instr_per_frame) from the counter passes of tools/profile.sh: FETCH_SIZE (KiB) x 1024 x 2 (the
gfx950 half-count correction of MI355X_MICROARCH.md's HBM recipe) and WRITE_SIZE (KiB) x 1024 per launch, SQ_INSTS_VALU (mean per
shader engine) x 32, over the 6144 streams x 1000 frames x 1200 samples each launch of that pass demodulates."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
STREAMS, FRAMES, N = 6144, 1000, 1200
vals = {}
kernel = None
for ln in open(src):
    m = re.match(r"(\S.*?)(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU)\s+(\d+)\s+([0-9.]+)", ln)
    if m:
        vals[m.group(2)] = float(m.group(4))
samples = STREAMS * FRAMES * N
sys.path.insert(0, ROOT)
out = {
    "kernel_name": None,
    "source": os.path.relpath(src, ROOT),
    "hbm_read_bytes_per_sample": vals["FETCH_SIZE"] * 1024 * 2 / samples,
    "hbm_write_bytes_per_sample": vals["WRITE_SIZE"] * 1024 / samples,
    "note": "FETCH_SIZE(KiB)x1024x2 (gfx950 half-count correction) + WRITE_SIZE(KiB)x1024, separate --pmc passes, per launch of "
            "6144 streams x 1.2e6 samples, packed bits written in place (the bench's output mode at every N)",
    "valu_instr_per_frame": round(vals["SQ_INSTS_VALU"] * 32 / (STREAMS * FRAMES)),
    "valu_note": "SQ_INSTS_VALU (mean per shader engine x 32) per stream-frame of 1200 samples; peak issue in the counters' unit = "
                 "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614.4 G wave-instr/s",
}
# the instance name as pirip_hip_get_kernel_name prints it (bench.py quotes the traffic only when its handle reports the same)
for ln in open(src.replace("_pmc.txt", "_stats.txt")):       # the bench line under the profiler, in the kernel-trace file of the same tag
    m = re.search(r'"kernel": "([^"]+)"', ln)
    if m:
        out["kernel_name"] = m.group(1)
        break
if out["kernel_name"] is None:
    del out["kernel_name"]
# the kernel build these counters belong to (bench.py quotes them only while the running library reports the same hash); the pmc
# file's header carries it when tools/profile.sh wrote it, else the library in the tree is asked
m = re.search(r"kernel_source_hash ([0-9a-f]{16})", open(src).read())
if m:
    out["kernel_source_hash"] = m.group(1)
else:
    import ctypes
    import pirip_amd
    L = pirip_amd.lib()
    L.pirip_hip_kernel_source_hash.restype = ctypes.c_char_p
    out["kernel_source_hash"] = L.pirip_hip_kernel_source_hash().decode()
# the opt-in band-only estimator's counter passes of the same tag (tools/profile.sh part "band"), when they were taken
bsrc = src.replace("_headline_pmc.txt", "_band_only_stats_pmc.txt")
if bsrc != src and os.path.exists(bsrc):
    bv = {}
    for ln in open(bsrc):
        m = re.match(r"(wave\S*?)(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU)\s+(\d+)\s+([0-9.]+)", ln)
        if m:
            bv[m.group(2)] = float(m.group(4))
    if "SQ_INSTS_VALU" in bv:
        out["band_only"] = {"source": os.path.relpath(bsrc, ROOT), "valu_instr_per_frame": round(bv["SQ_INSTS_VALU"] * 32 / (STREAMS * FRAMES)),
                            "hbm_read_bytes_per_sample": bv.get("FETCH_SIZE", 0.0) * 1024 * 2 / samples,
                            "hbm_write_bytes_per_sample": bv.get("WRITE_SIZE", 0.0) * 1024 / samples}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
