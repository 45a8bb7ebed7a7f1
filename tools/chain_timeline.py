#!/usr/bin/env python3
"""tools/chain_timeline.py <rocprofv3 output dir> -- the kernels of the LAST config-4 chain call of a `rocprofv3 --kernel-trace -- python
tools/chain_ab.py --child 3.5` run as a timeline: start and end relative to the call's first kernel (ms), which kernels overlap.
(A chain call = everything between two consecutive hist_prepare launches of the first stream range.)"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
    c = sqlite3.connect(db)
    cols = [x[1] for x in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(c.execute(f"select name, start, end, grid_x{', ' + qcol if qcol else ''} from kernels order by start"))
    wanted = ("wave<", "fsk_demod_wave", "decode_", "uwbest", "fsm_kernel", "hist_prepare", "save_hist", "llr_tile")
    rows = [r for r in rows if any(w in r[0] for w in wanted)]
    # the last run of kernels that contains no llr_tile (the stand-alone receive stage's marker) and starts with hist_prepare
    idx = [i for i, r in enumerate(rows) if "hist_prepare" in r[0]]
    if len(idx) < 2:
        print("no chain call found"); return
    # calls come as pairs of hist_prepare (two ranges); take the last pair's span up to the next llr_tile / end
    start_i = idx[-2]
    end_i = len(rows)
    for i in range(start_i, len(rows)):
        if "llr_tile" in rows[i][0]:
            end_i = i; break
    t0 = rows[start_i][1]
    print(f"# {os.path.basename(db)}: kernels of the last chain call, ms from its first kernel")
    for r in rows[start_i:end_i]:
        name = r[0].replace("void pirip::fsk_demod_wave_kernel", "wave").replace("(anonymous namespace)::", "").split("(")[0][:48]
        print(f"{name:<50} {(r[1] - t0) / 1e6:8.3f} -> {(r[2] - t0) / 1e6:8.3f}  ({(r[2] - r[1]) / 1e6:7.3f} ms)  grid {r[3]}" + (f"  queue {r[4]}" if qcol else ""))


if __name__ == "__main__":
    main()
