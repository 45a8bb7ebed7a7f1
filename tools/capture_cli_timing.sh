#!/bin/bash
# tools/capture_cli_timing.sh -- fsk_demod on a 100 M-sample u8 file (the capture route) against the same bytes through a pipe (the read loop,
# 4096 frames per GPU call): wall time of the whole process, and that the outputs are identical. Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$R/pirip_amd/bin
T=${TMPDIR:-/tmp}/capcli
mkdir -p $T
NB=${1:-4166000}     # bits: x 24 samples
$B/fsk_get_test_bits - $NB | $B/fsk_mod -c -a 8000 2 240000 10000 10000 10000 - - > $T/s16.raw 2>/dev/null
python3 - $T/s16.raw $T/in.u8 <<'PY'
import sys, numpy as np
x = np.fromfile(sys.argv[1], dtype=np.int16).astype(np.float32) / 8000.0 * 2.0
rng = np.random.default_rng(1)
x += rng.normal(0.0, 0.9, x.shape).astype(np.float32)          # Eb/N0 about 18 dB
np.clip(np.rint(127.0 + 20.0 * x), 0, 255).astype(np.uint8)[14:].tofile(sys.argv[2])
PY
ls -l $T/in.u8 | awk '{print "# input bytes:", $5}'
now() { date +%s.%N; }
for i in 1 2; do
  t0=$(now)
  PIRIP_FSK_DEMOD_REPORT=1 $B/fsk_demod -d -p 24 2 240000 10000 $T/in.u8 $T/out_file.bits 2>&1 | grep -v Setting
  echo "# file route  (fsk_demod -d -p 24 2 240000 10000 in.u8 out): $(python3 -c "print(round($(now) - $t0, 3))") s wall"
done
t0=$(now)
cat $T/in.u8 | $B/fsk_demod -d -p 24 2 240000 10000 - - 2>/dev/null > $T/out_pipe.bits
echo "# pipe route  (cat in.u8 | fsk_demod ... - -): $(python3 -c "print(round($(now) - $t0, 3))") s wall"
cmp $T/out_file.bits $T/out_pipe.bits && echo "# outputs identical: $(stat -c %s $T/out_file.bits) bytes"
$B/fsk_put_test_bits - < $T/out_file.bits 2>&1 | tail -2
rm -rf $T
