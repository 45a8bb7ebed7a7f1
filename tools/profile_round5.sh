#!/bin/bash
# tools/profile_round5.sh TAG -- the rocprofv3 passes behind profiles/r05_*: run on the GPU box
#   gpurun -- 'bash tools/profile_round5.sh r05_a'
# 1. headline (bench.py): kernel-trace stats + PMC passes (each counter set its own run, --kernel-trace only; FETCH_SIZE and
#    WRITE_SIZE in separate passes, MI355X_MICROARCH.md's HBM recipe)
# 2. the other kernels through tools/bench_configs.py: config 4 (4-FSK wave instance, bits out), config 4 whole chain (the fused
#    hand-over: wave instance writing bit LLRs, unique-word search, state machine, fast decoder), config 3 (decimator + Ts = 40 instance)
tag=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
# PARTS: which sections to run (default all): headline band configs_stats configs_pmc. Every rocprofv3 run has its own short timeout:
# on 2026-09-29 one box hung in a --pmc pass and the old 600 / 900 s limits let a single call burn 40 GPU-minutes.
PARTS=${PARTS:-"headline band configs_stats configs_pmc"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
SETS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC")
if has headline; then
O=$R/gpurun_out/${tag}_headline_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16" > $O
rm -rf /tmp/pr; timeout ${STEP_TIMEOUT:-240} rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16 > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr | head -6 >> $O
echo "# bench line under the profiler:" >> $O; grep '^{' /tmp/pr.log | cut -c1-1600 >> $O
O=$R/gpurun_out/${tag}_headline_pmc.txt
echo "# PMC passes, each its own run of: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4" > $O
echo "# kernel_source_hash $(python3 -c "import sys,ctypes; sys.path.insert(0,'$R'); import pirip_amd; L=pirip_amd.lib(); L.pirip_hip_kernel_source_hash.restype=ctypes.c_char_p; print(L.pirip_hip_kernel_source_hash().decode())")" >> $O
echo "# (6144 streams = two rounds of the 3072 resident waves; packed-bit output, the bench's mode at every N; values are means per shader engine (x32 for the chip) except FETCH/WRITE_SIZE (KiB, chip))" >> $O
for set in "${SETS[@]}"; do
  rm -rf /tmp/pm; timeout ${STEP_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4 > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm fsk_demod | cut -c1-24,52-140 >> $O
done
fi
if has band; then
# 1b. the same workload through the OPT-IN band-only estimator (PIRIP_EST_BAND=1: pirip_hip_set_estimator_band_only where it applies)
O=$R/gpurun_out/${tag}_band_only_stats_pmc.txt
echo "# PIRIP_EST_BAND=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16" > $O
rm -rf /tmp/pr; PIRIP_EST_BAND=1 timeout ${STEP_TIMEOUT:-240} rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16 > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr | head -6 >> $O
echo "# bench line under the profiler:" >> $O; grep '^{' /tmp/pr.log | cut -c1-1200 >> $O
echo "# PMC passes (6144 streams), each its own run: FETCH_SIZE, WRITE_SIZE, instruction counts" >> $O
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pm; PIRIP_EST_BAND=1 timeout ${STEP_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4 > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm fsk_demod | cut -c1-24,52-140 >> $O
done
fi
if has configs_stats; then
O=$R/gpurun_out/${tag}_configs_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py --iters 10" > $O
rm -rf /tmp/pr; timeout ${CFG_TIMEOUT:-360} rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/bench_configs.py --iters 10 > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr | head -16 >> $O
echo "# tool output under the profiler:" >> $O; grep '^{' /tmp/pr.log >> $O
fi
if has configs_pmc; then
O=$R/gpurun_out/${tag}_configs_pmc.txt
echo "# PMC passes over tools/bench_configs.py --iters 2 (config 4: 8192 x 600k samples 4-FSK, bits out / fused FSK_LDPC chain at 7 and 3.5 dB; config 3: 4096 x 1.8e6 u8 -> /45 -> demod); per kernel, means per shader engine" > $O
for set in "${SETS[@]}"; do
  rm -rf /tmp/pm; timeout ${CFG_TIMEOUT:-360} rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/bench_configs.py --iters 2 > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm "_kernel" | grep -v "synth\|elementwise\|at::\|vectorized" | cut -c1-48,52-140 >> $O
done
fi
