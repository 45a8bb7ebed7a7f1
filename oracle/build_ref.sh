#!/bin/bash
# oracle/build_ref.sh <codec2-checkout> <csdr-checkout>   --   the one-command exit from "parity unpinned".
#
# TEST INFRASTRUCTURE ONLY. Builds UPSTREAM's own implementation of the hot path -- the sources pirip clones with
# /root/reference/build_codec2.sh:3-5 and build_csdr.sh:4-5 -- from where they lie, with plain cc (no cmake, no stand-in
# headers or generated files: if a source needs one, this script stops and says which), outputs ONLY into oracle/_ref/
# (git-ignored, not gpurun-ignored), then runs oracle/pin_against_ref.py, which feeds the inputs of the committed
# fixtures (tests/golden/*.npz) and the 600 000-bit config-1 vector through upstream's binaries and this repo's oracle
# and writes the comparison to tests/golden/PINNED.json. With that file saying "pinned": true, DESIGN.md section 2's
# "parity unpinned" label can be dropped for the rows it lists.
#
# It cannot run in the build container: neither checkout exists there and there is no network.
# The source file lists below are [UPSTREAM-RECALLED] (codec2 ~v1.0, csdr master); the script checks each file exists
# and reports the first that does not instead of guessing.
set -u
C2=${1:-}; CSDR=${2:-}
if [ -z "$C2" ] || [ -z "$CSDR" ] || [ ! -d "$C2/src" ] || [ ! -f "$CSDR/libcsdr.c" ]; then
    echo "usage: $0 <codec2-checkout (has src/fsk.c)> <csdr-checkout (has libcsdr.c)>" >&2
    exit 2
fi
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
mkdir -p "$OUT"
CC=${CC:-cc}
# what an x86-64 baseline build of upstream computes: no fused multiply-add, no fast-math
CFLAGS="-O2 -std=gnu11 -ffp-contract=off -fno-fast-math -I$C2/src"

need() { for f in "$@"; do [ -f "$f" ] || { echo "build_ref: missing $f (file list is recalled -- adjust it to this checkout)" >&2; exit 3; }; done; }
build() {   # build <output> <sources...>
    local out=$1; shift
    need "$@"
    echo "cc -> $out"
    $CC $CFLAGS -o "$OUT/$out" "$@" -lm || { echo "build_ref: $out does not build from its own sources with plain cc (a generated header such as version.h, or a library, is needed): treat as unbuildable, do NOT write a stand-in" >&2; exit 4; }
}

S=$C2/src
# modem core + its tools (codec2 src/CMakeLists.txt builds these from the same files)
FSK_CORE="$S/fsk.c $S/kiss_fft.c $S/kiss_fftr.c $S/modem_probe.c $S/modem_stats.c $S/octave.c"
build fsk_demod          $S/fsk_demod.c $FSK_CORE
build fsk_mod            $S/fsk_mod.c $FSK_CORE
build fsk_get_test_bits  $S/fsk_get_test_bits.c
build fsk_put_test_bits  $S/fsk_put_test_bits.c
# LDPC decoder on its own: ldpc_dec reads soft decisions / LLRs and decodes with a named code
LDPC_SRCS="$S/ldpc_dec.c $S/mpdecode_core.c $S/ldpc_codes.c $S/phi0.c $S/ofdm.c $S/interldpc.c $S/gp_interleaver.c $S/filter.c $S/kiss_fft.c $S/kiss_fftr.c $S/modem_stats.c $S/modem_probe.c $S/octave.c"
build ldpc_dec           $LDPC_SRCS $(ls $S/H*.c 2>/dev/null)
# csdr: the command-line tool links libcsdr.c + libcsdr_gpl.c + fft backends; only three functions are on the path, so a
# 40-line driver (oracle/ref_csdr_main.c, this repo's, calling upstream's functions by their own names) is linked instead
need "$CSDR/libcsdr.c" "$CSDR/libcsdr.h"
# TWICE: csdr_path is the strict scalar loop (what oracle/csdr_oracle.c and the product's default tap loop
# state); csdr_path_fast is built with upstream's OWN flags [UPSTREAM-RECALLED csdr Makefile: -O3 -ffast-math, on x86 plus
# -march=native style vector flags], i.e. the summation order the shipped binary's vectoriser picks on THIS machine.
# pin_against_ref.py records both, so that it is known which tap-loop arithmetic of the product (pirip_hip_decim_set_arith)
# the real binary agrees with.
echo "cc -> csdr_path"
$CC -O2 -std=gnu11 -ffp-contract=off -fno-fast-math -I"$CSDR" -DLIBCSDR_GPL=0 -o "$OUT/csdr_path" "$HERE/ref_csdr_main.c" "$CSDR/libcsdr.c" -lm \
    || { echo "build_ref: libcsdr.c does not build alone (it may want fftw3 / libcsdr_gpl.c): see the note in ref_csdr_main.c" >&2; exit 4; }
echo "cc -> csdr_path_fast"
$CC -O3 -ffast-math -std=gnu11 -I"$CSDR" -DLIBCSDR_GPL=0 -o "$OUT/csdr_path_fast" "$HERE/ref_csdr_main.c" "$CSDR/libcsdr.c" -lm \
    || echo "build_ref: the -O3 -ffast-math build of libcsdr failed; PINNED.json will carry the strict build only" >&2
# the FSK_LDPC code table for the product: H, unique word and thresholds in the code-file format of pirip_amd/csrc/fsk_ldpc.hpp
python3 "$HERE/extract_code_table.py" "$C2" H_256_512_4 > "$OUT/H_256_512_4.code" || echo "build_ref: code table extraction failed (see message); the LDPC rows stay unpinned" >&2
( cd "$C2" && git rev-parse HEAD 2>/dev/null ) > "$OUT/codec2.commit"
( cd "$CSDR" && git rev-parse HEAD 2>/dev/null ) > "$OUT/csdr.commit"
exec python3 "$HERE/pin_against_ref.py" "$OUT"
