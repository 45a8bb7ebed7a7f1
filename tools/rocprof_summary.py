#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as text:
per-kernel calls / avg / min / max / total duration, plus launch geometry and register counts.
usage: tools/rocprof_summary.py <dir-or-db> [> profiles/rNN_xxx.txt]"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db in dbs:
        c = sqlite3.connect(db)
        print(f"# {os.path.basename(db)}")
        print(f"{'kernel':<60} {'calls':>6} {'avg_ms':>10} {'min_ms':>10} {'max_ms':>10} {'total_ms':>10} {'%':>6}  grid wg lds vgpr sgpr scratch")
        rows = list(c.execute(
            "select name, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, sum(end-start)/1e6,"
            " max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count), max(scratch_size)"
            " from kernels group by name order by 6 desc"))
        tot = sum(r[5] for r in rows) or 1.0
        for r in rows[:25]:
            name = r[0].replace("void pirip::fsk_demod_wave_kernel", "wave").replace(", ", ",")
            name = name.split("(")[0] if name.startswith("wave<") else name
            name = name if len(name) <= 58 else name[:55] + "..."
            print(f"{name:<60} {r[1]:>6} {r[2]:>10.4f} {r[3]:>10.4f} {r[4]:>10.4f} {r[5]:>10.3f} {100 * r[5] / tot:>6.2f}  "
                  f"{r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]}")
        try:
            pmc = list(c.execute("select name from sqlite_master where name like '%pmc%'"))
            if pmc:
                print("# pmc tables:", [p[0] for p in pmc])
        except Exception:
            pass


if __name__ == "__main__":
    main()
