#!/usr/bin/env python3
"""tools/make_standin_code.py -- generate pirip_amd/data/standin_256_512_4.code, a STAND-IN for codec2's H_256_512_4.

The reference names the code (`--code H_256_512_4`, /root/reference/README.md:184, script/frame_repeater:8) but its
parity-check matrix, the 32-bit unique word and the sync thresholds live in codec2, which is not in /root/reference
(SURVEY.md 7.6). Nothing here is a reconstruction of those tables: it is a seeded construction of a code with the same
SHAPE -- rate 1/2, 256 data + 256 parity bits, data columns of weight 4, repeat-accumulate (dual-diagonal) parity part so
that upstream's linear-time encoder applies -- so that the decoder, framer and receiver can be built and tested. The
unique word is the 32-bit CCSDS attached sync marker 0x1ACFFC1D (a public constant with low autocorrelation sidelobes),
again a stand-in. To use codec2's real tables, write them in the same file format (pirip_amd/csrc/fsk_ldpc.hpp).

Construction: every data column gets 4 rows, chosen greedily from the least-loaded rows at random (seed 2026), such that
no two data columns share two rows (no 4-cycles inside H1) and no data column sits in two adjacent rows (no 4-cycles
through the accumulator's dual diagonal); rows end up with 4 data ones each."""
import os
import random
import sys

K, M_PAR, WCOL, SEED = 256, 256, 4, 2026
UW = 0x1ACFFC1D


def build():
    rng = random.Random(SEED)
    for attempt in range(200):
        rows_of = []
        load = [0] * M_PAR
        pair_used = set()
        ok = True
        for c in range(K):
            placed = None
            for _ in range(400):
                cand = sorted(range(M_PAR), key=lambda r: (load[r], rng.random()))[:24]
                pick = []
                for r in cand:
                    if any(abs(r - q) <= 1 or (min(r, q), max(r, q)) in pair_used for q in pick):
                        continue
                    pick.append(r)
                    if len(pick) == WCOL:
                        break
                if len(pick) == WCOL:
                    placed = pick
                    break
                rng.random()
            if placed is None:
                ok = False
                break
            for i in range(WCOL):
                for j in range(i + 1, WCOL):
                    pair_used.add((min(placed[i], placed[j]), max(placed[i], placed[j])))
            for r in placed:
                load[r] += 1
            rows_of.append(placed)
        if ok and max(load) - min(load) <= 1:
            return rows_of
    raise SystemExit("construction failed")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "pirip_amd", "data", "standin_256_512_4.code")
    rows_of = build()
    rows = [[] for _ in range(M_PAR)]
    for c, rs in enumerate(rows_of):
        for r in rs:
            rows[r].append(c)
    for p in range(M_PAR):
        if p:
            rows[p].append(K + p - 1)
        rows[p].append(K + p)
    with open(out, "w") as f:
        f.write("# pirip_hip LDPC code file v1 -- STAND-IN, not codec2's H_256_512_4 (see tools/make_standin_code.py)\n")
        f.write("name STANDIN_256_512_4\n")
        f.write(f"n {K + M_PAR}\nk {K}\nmax_iter 15\n")
        f.write("uw " + " ".join(str((UW >> (31 - i)) & 1) for i in range(32)) + "\n")
        f.write("uw_thresh1 5\nuw_thresh2 6\nbad_uw_thresh 1\n")
        f.write(f"rows {M_PAR}\n")
        for r in rows:
            f.write(" ".join(str(c) for c in sorted(r)) + "\n")
    print("wrote", out, "edges", sum(len(r) for r in rows))


if __name__ == "__main__":
    main()
