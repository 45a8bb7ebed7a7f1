#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc results (rocpd sqlite) per kernel and counter: mean per dispatch.
usage: tools/pmc_extract.py <dir-or-db> [kernel-substring]"""
import glob
import os
import sqlite3
import sys


def short(name):
    """wave-kernel instances differ only in their template arguments: keep those readable inside the column width"""
    n = name.replace("void pirip::fsk_demod_wave_kernel", "wave").replace("(anonymous namespace)::", "").replace(", ", ",")
    return n.split("(")[0] if n.startswith("wave<") else n


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "fsk_demod"
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [x[1] for x in c.execute("pragma table_info(pmc_events)")]
        # typical columns: ... name (kernel), counter_name / pmc name, value / counter_value, dispatch_id
        namecol = "name" if "name" in cols else [x for x in cols if "kernel" in x][0]
        ccol = [x for x in cols if x in ("counter_name", "pmc_name", "counter")]
        vcol = [x for x in cols if x in ("value", "counter_value")]
        if not ccol or not vcol:
            print("# columns:", cols)
            for r in c.execute("select * from pmc_events limit 3"):
                print(r)
            continue
        q = (f"select {namecol}, {ccol[0]}, count(*), avg({vcol[0]}), min({vcol[0]}), max({vcol[0]}) from pmc_events "
             f"where {namecol} like ? group by {namecol}, {ccol[0]} order by 1, 2")
        print(f"# {os.path.basename(db)}: per-dispatch counter values (kernels matching '{pat}')")
        print(f"{'kernel':<50} {'counter':<26} {'n':>4} {'mean':>18} {'min':>18} {'max':>18}")
        for r in c.execute(q, (f"%{pat}%",)):
            k = short(r[0])
            k = k if len(k) <= 48 else k[:45] + "..."
            print(f"{k:<50} {r[1]:<26} {r[2]:>4} {r[3]:>18.1f} {r[4]:>18.1f} {r[5]:>18.1f}")


if __name__ == "__main__":
    main()
