#!/bin/bash
# tools/phase_valu.sh [INSTANCE ...] -- instructions each phase of the wave demodulator EXECUTES per frame.
#   gpurun -- 'bash tools/phase_valu.sh > gpurun_out/r04_phase_valu.txt'
# A -DPIRIP_WAVE_TIMING library ends every frame after phase k when PIRIP_WAVE_STOP=k (a run-time test, so the code before the
# mark is the code of the full frame); one rocprofv3 --pmc pass per k (counters only, --kernel-trace); the per-phase figure is the
# difference between consecutive k. Counter values are means per shader engine (x 32 for the chip).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
make -s -C $R/pirip_amd/csrc -j16 LIBDIR=../lib_timing EXTRA=-DPIRIP_WAVE_TIMING ../lib_timing/libpirip_hip.so >/dev/null 2>&1 || { echo "timing build failed"; exit 1; }
export PIRIP_HIP_LIB=$R/pirip_amd/lib_timing/libpirip_hip.so
INSTS=${@:-headline_2fsk_p24_u8d 4fsk_p8_u8d}
echo "# tools/phase_valu.sh: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS per 1200-sample frame, cumulative up to the mark and per phase"
echo "# (timing-build kernel: the s_memtime marks and the stop test add ~10 scalar instructions per frame; VALU is unaffected)"
for inst in $INSTS; do
  python3 - "$inst" <<'PY' > /tmp/pv_hdr.txt
import sys
print("## " + sys.argv[1])
PY
  cat /tmp/pv_hdr.txt
  prev_v=0; prev_s=0; prev_l=0
  for k in 1 2 3 4 none; do
    rm -rf /tmp/pv
    if [ $k = none ]; then unset PIRIP_WAVE_STOP; else export PIRIP_WAVE_STOP=$k; fi
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pv -- python3 $R/tools/phase_valu.py $inst > /tmp/pv.log 2>&1
    frames=$(grep PHASE_VALU /tmp/pv.log | sed 's/.*frames_total \([0-9]*\).*/\1/')
    python3 $R/tools/pmc_extract.py /tmp/pv fsk_demod > /tmp/pv_pmc.txt
    python3 - "$k" "$frames" <<'PY'
import sys
k, frames = sys.argv[1], float(sys.argv[2])
names = {"1": "estimator FFTs", "2": "+ peak pick", "3": "+ correlator", "4": "+ DMA issue, hist copy, window sums, timing", "none": "+ atan2, resample, decide, outputs (whole frame)"}
v = {}
for ln in open("/tmp/pv_pmc.txt"):
    f = ln.split()
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
        if c in f:
            v[c] = float(f[f.index(c) + 2])
print(f"stop {k:>4s} {names[k]:52s} frames {int(frames):9d}  VALU {v['SQ_INSTS_VALU'] * 32 / frames:8.1f}  SALU {v['SQ_INSTS_SALU'] * 32 / frames:7.1f}  "
      f"LDS {v['SQ_INSTS_LDS'] * 32 / frames:6.1f}  (waves {v['SQ_WAVES'] * 32:.0f})")
PY
  done
done
