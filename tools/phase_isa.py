#!/usr/bin/env python3
"""Per-phase instruction mix of the wave demodulator: splits the frame loop of an instrumented build
(-DPIRIP_WAVE_TIMING, `hipcc --save-temps`) at its s_memtime marks and counts instruction classes in program order
(static counts: loops with `#pragma unroll 1` appear once -- the FFT batch loop runs NFFT/4 times per frame).

    python tools/phase_isa.py /tmp/ti/fsk_demod_wave-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys
from collections import Counter

TRANS = ("v_sqrt", "v_rsq", "v_rcp", "v_exp", "v_log", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.endswith("_dpp") or op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "valu_dpp/lane"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu_plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().splitlines()
    seg, segs, nops = Counter(), [], 0
    inside = False
    for ln in lines:
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", ln)
        if not m or ln.lstrip().startswith((".", ";")):
            continue
        op, rest = m.group(1), m.group(2)
        if op == "s_memtime":
            segs.append(seg)
            seg = Counter()
            continue
        c = classify(op)
        seg[c] += 1
        if op == "s_nop":
            seg["nop_cycles"] += int(rest.split()[0]) + 1
    segs.append(seg)
    keys = ["valu_plain", "valu_packed", "valu_dpp/lane", "valu_trans", "lds", "vmem", "salu", "branch", "s_waitcnt", "s_nop", "nop_cycles"]
    print("segment " + " ".join(f"{k:>13s}" for k in keys))
    for i, s in enumerate(segs):
        print(f"{i:7d} " + " ".join(f"{s.get(k, 0):13d}" for k in keys))


if __name__ == "__main__":
    main()
