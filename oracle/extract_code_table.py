#!/usr/bin/env python3
"""oracle/extract_code_table.py <codec2-checkout> <CODE_NAME>  ->  code file (pirip_amd/csrc/fsk_ldpc.hpp format) on stdout.

TEST / INTEGRATION INFRASTRUCTURE (run by oracle/build_ref.sh): turns codec2's own LDPC table and FSK_LDPC framing constants
into the DATA the product loads with `--code` -- the product contains no code table of its own. Everything about
upstream's file layout here is [UPSTREAM-RECALLED] and checked while parsing; a mismatch stops with a message rather than
emitting a guess:
  src/<NAME>.h   #define <NAME>_NUMBERPARITYBITS m, _MAX_ROW_WEIGHT, _CODELENGTH n, _NUMBERROWSHCOLS k, _MAX_COL_WEIGHT
  src/<NAME>.c   <NAME>_H_rows[m * max_row_weight]: column-major, entry [r + j*m] = 1-based data column of check r, 0 = none
                 (repeat-accumulate codes: the parity part of H is the implicit dual diagonal, columns k+r and k+r-1)
  src/freedv_fsk.c / freedv_api.c   the 32-bit unique word and the sync thresholds of FSK_LDPC mode
"""
import re
import sys


def die(msg):
    sys.stderr.write("extract_code_table: " + msg + "\n")
    sys.exit(1)


def main():
    if len(sys.argv) != 3:
        die(__doc__)
    c2, name = sys.argv[1], sys.argv[2]
    try:
        hdr = open(f"{c2}/src/{name}.h").read()
        src = open(f"{c2}/src/{name}.c").read()
    except OSError as e:
        die(str(e))

    def macro(suffix):
        m = re.search(r"#define\s+%s_%s\s+(\d+)" % (name, suffix), hdr + src)
        if not m:
            die(f"{name}_{suffix} not found")
        return int(m.group(1))
    m_par, n, k, wr = macro("NUMBERPARITYBITS"), macro("CODELENGTH"), macro("NUMBERROWSHCOLS"), macro("MAX_ROW_WEIGHT")
    if n != k + m_par:
        die(f"n = {n}, k = {k}, m = {m_par}: not a systematic code with k = NUMBERROWSHCOLS")
    arr = re.search(r"%s_H_rows\s*\[\s*\]\s*=\s*\{([^}]*)\}" % name, src + hdr)
    if not arr:
        die("H_rows array not found")
    vals = [int(v) for v in re.findall(r"\d+", arr.group(1))]
    if len(vals) != m_par * wr:
        die(f"H_rows has {len(vals)} entries, expected {m_par} x {wr}")
    rows = []
    for r in range(m_par):
        cols = sorted(vals[r + j * m_par] - 1 for j in range(wr) if vals[r + j * m_par] > 0)
        if any(c < 0 or c >= k for c in cols):
            die(f"row {r}: data column out of range (is H_rows 1-based and data-only?)")
        cols += ([k + r - 1] if r else []) + [k + r]
        rows.append(sorted(cols))
    text = ""
    for f in ("freedv_fsk.c", "freedv_api.c", "freedv_api_internal.h"):
        try:
            text += open(f"{c2}/src/{f}").read()
        except OSError:
            pass
    uw = re.search(r"fsk_ldpc_uw\s*\[\s*\]\s*=\s*\{([^}]*)\}", text)
    if not uw:
        die("FSK_LDPC unique word (fsk_ldpc_uw[]) not found in src/freedv_fsk.c / freedv_api.c")
    uwb = [int(v) for v in re.findall(r"[01]", uw.group(1))]
    if len(uwb) != 32:
        die(f"unique word has {len(uwb)} bits, expected 32")

    def thresh(var, default):
        mm = re.search(r"%s\s*=\s*(\d+)" % var, text)
        return int(mm.group(1)) if mm else default
    out = [f"# {name}: extracted from codec2 by oracle/extract_code_table.py -- codec2's table, not a stand-in",
           f"name {name}", f"n {n}", f"k {k}", "max_iter 15", "uw " + " ".join(map(str, uwb)),
           f"uw_thresh1 {thresh('fsk_ldpc_thresh1', 5)}", f"uw_thresh2 {thresh('fsk_ldpc_thresh2', 6)}",
           f"bad_uw_thresh {thresh('fsk_ldpc_baduw_thresh', 1)}", f"rows {m_par}"]
    out += [" ".join(map(str, r)) for r in rows]
    print("\n".join(out))


if __name__ == "__main__":
    main()
