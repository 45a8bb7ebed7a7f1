#!/bin/bash
# tools/profile.sh TAG [PART ...] -- every rocprofv3 pass behind profiles/<TAG>_*: run on the GPU box
#   gpurun -- 'bash tools/profile.sh r06_a headline configs_stats'
# (replaces profile_round{2,3,4,5}.sh, profile_headline / _configs / _block / _instances.sh, pmc_decim / pmc_decode / pmc_fast_extra.sh.)
# Summaries go to gpurun_out/<TAG>_<part>_{stats,pmc}.txt; copy the ones to be judged into profiles/. Counters are collected as
# MI355X_MICROARCH.md prescribes: each set in its own run, --kernel-trace --pmc only, FETCH_SIZE and WRITE_SIZE in separate passes.
# Every rocprofv3 run has its own short timeout (STEP_TIMEOUT / CFG_TIMEOUT seconds): a box that hangs in a --pmc pass must not
# burn the call's whole limit.
# PARTS (default: headline band configs_stats configs_pmc):
#   headline       bench.py (BASELINE configs[1]): kernel-trace stats at the default shape; counter sets at 6144 streams
#   band           the same through the OPT-IN band-only estimator (PIRIP_EST_BAND=1): stats + FETCH / WRITE / instruction counts
#   configs_stats  tools/bench_configs.py (config 4 demod and whole chain, config 3, the rtl_fsk shapes): kernel-trace stats
#   configs_pmc    ... its counter sets, per kernel
#   decode         the FSK_LDPC decoder kernels on the config-4 chain workload (tools/chain_ab.py's child), once per
#                  PIRIP_LDPC_DECODER in $DECODERS (default "fast bank"): stats + the LDS / VALU counter sets
#   block          the Ts = 240 block instance through tools/instance_rates.py (6144 streams): counter sets
#   instances      every wave-kernel instance family through tools/instance_rates.py: stats + counter sets
#   decim          tools/bench_decim.py (csdr decimator): stats + FETCH / WRITE
tag=${1:-r06_x}; shift
PARTS="${*:-headline band configs_stats configs_pmc}"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$(dirname $R/gpurun_out/${tag}_x)"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
ST=${STEP_TIMEOUT:-240}; CT=${CFG_TIMEOUT:-360}
SQ1="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ3="SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
SQ4="SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
SETS=("FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2" "$SQ3")
khash() { python3 -c "import sys,ctypes; sys.path.insert(0,'$R'); import pirip_amd; L=pirip_amd.lib(); L.pirip_hip_kernel_source_hash.restype=ctypes.c_char_p; print(L.pirip_hip_kernel_source_hash().decode())"; }
# stats CMD... : one kernel-trace --stats run of CMD into /tmp/pr (+ /tmp/pr.log)
stats() { rm -rf /tmp/pr; timeout $1 rocprofv3 --kernel-trace --stats -d /tmp/pr -- "${@:2}" > /tmp/pr.log 2>&1; }
# pmc TIMEOUT "SET" CMD... : one counter run of CMD into /tmp/pm
pmc() { rm -rf /tmp/pm; timeout $1 rocprofv3 --kernel-trace --pmc $2 -d /tmp/pm -- "${@:3}" > /tmp/pm.log 2>&1; }

if has headline; then
  O=$R/gpurun_out/${tag}_headline_stats.txt
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16" > $O
  stats $ST python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16
  python $R/tools/rocprof_summary.py /tmp/pr | head -6 >> $O
  echo "# bench line under the profiler:" >> $O; grep '^{' /tmp/pr.log | cut -c1-1600 >> $O
  O=$R/gpurun_out/${tag}_headline_pmc.txt
  echo "# PMC passes, each its own run of: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4" > $O
  echo "# kernel_source_hash $(khash)" >> $O
  echo "# (6144 streams = two rounds of the 3072 resident waves; packed-bit output, the bench's mode at every N; values are means per shader engine (x32 for the chip) except FETCH/WRITE_SIZE (KiB, chip))" >> $O
  for set in "${SETS[@]}"; do
    pmc $ST "$set" python $R/bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4
    python $R/tools/pmc_extract.py /tmp/pm fsk_demod | cut -c1-24,52-140 >> $O
  done
fi
if has band; then
  O=$R/gpurun_out/${tag}_band_only_stats_pmc.txt
  echo "# PIRIP_EST_BAND=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16" > $O
  PIRIP_EST_BAND=1 stats $ST python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --check-streams 16
  python $R/tools/rocprof_summary.py /tmp/pr | head -6 >> $O
  echo "# bench line under the profiler:" >> $O; grep '^{' /tmp/pr.log | cut -c1-1200 >> $O
  echo "# PMC passes (6144 streams), each its own run: FETCH_SIZE, WRITE_SIZE, instruction counts" >> $O
  for set in "FETCH_SIZE" "WRITE_SIZE" "$SQ1"; do
    PIRIP_EST_BAND=1 pmc $ST "$set" python $R/bench.py --streams 6144 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --check-streams 4
    python $R/tools/pmc_extract.py /tmp/pm fsk_demod | cut -c1-24,52-140 >> $O
  done
fi
if has configs_stats; then
  O=$R/gpurun_out/${tag}_configs_stats.txt
  echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py --iters 10" > $O
  stats $CT python $R/tools/bench_configs.py --iters 10
  python $R/tools/rocprof_summary.py /tmp/pr | head -16 >> $O
  echo "# tool output under the profiler:" >> $O; grep '^{' /tmp/pr.log >> $O
fi
if has configs_pmc; then
  O=$R/gpurun_out/${tag}_configs_pmc.txt
  echo "# PMC passes over tools/bench_configs.py --iters 2 (config 4: 8192 x 600k samples 4-FSK, bits out / fused FSK_LDPC chain at 7 and 3.5 dB; config 3: 4096 x 1.8e6 u8 -> /45 -> demod); per kernel, means per shader engine" > $O
  for set in "${SETS[@]}"; do
    pmc $CT "$set" python $R/tools/bench_configs.py --iters 2
    python $R/tools/pmc_extract.py /tmp/pm "_kernel" | grep -v "synth\|elementwise\|at::\|vectorized" | cut -c1-48,52-140 >> $O
  done
fi
if has decode; then
  ebno=${EBNO:-3.5}
  O=$R/gpurun_out/${tag}_decode_pmc.txt
  echo "# FSK_LDPC decoder kernels: python tools/chain_ab.py --child $ebno (8192 streams x 600k samples: whole chain + stand-alone receive stage), per PIRIP_LDPC_DECODER; counters are means per dispatch and shader engine" > $O
  for dec in ${DECODERS:-fast bank}; do
    echo "## PIRIP_LDPC_DECODER=$dec $EXTRA_ENV" >> $O
    env $EXTRA_ENV PIRIP_LDPC_DECODER=$dec AB_ITERS=4 bash -c "$(declare -f stats); stats 300 python $R/tools/chain_ab.py --child $ebno"
    python $R/tools/rocprof_summary.py /tmp/pr | head -12 >> $O
    grep ABCHAIN /tmp/pr.log >> $O
    for set in "$SQ1" "$SQ2" "$SQ3"; do
      env $EXTRA_ENV PIRIP_LDPC_DECODER=$dec AB_ITERS=2 bash -c "$(declare -f pmc); pmc 300 '$set' python $R/tools/chain_ab.py --child $ebno"
      python $R/tools/pmc_extract.py /tmp/pm "decode_" | cut -c1-48,52-140 >> $O
    done
  done
fi
if has block; then
  export PIRIP_RATES_BLOCK_STREAMS=6144 PIRIP_RATES_GENERAL_ONLY=1 PIRIP_RATES_BLOCK_ONLY=1
  O=$R/gpurun_out/${tag}_block_pmc.txt
  echo "# PMC passes over: PIRIP_RATES_BLOCK_STREAMS=6144 python tools/instance_rates.py (block instance only: 6144 streams x 24 frames of 12000 samples, 2-FSK peak and 4-FSK mask); means per shader engine" > $O
  for set in "${SETS[@]}" "$SQ4"; do
    pmc 600 "$set" python $R/tools/instance_rates.py
    python $R/tools/pmc_extract.py /tmp/pm "block" | cut -c1-102 >> $O
  done
  tail -3 /tmp/pm.log >> $O
  unset PIRIP_RATES_BLOCK_STREAMS PIRIP_RATES_GENERAL_ONLY PIRIP_RATES_BLOCK_ONLY
fi
if has instances; then
  export PIRIP_RATES_WAVE_ONLY=1
  O=$R/gpurun_out/${tag}_instances_stats.txt
  echo "# PIRIP_RATES_WAVE_ONLY=1 rocprofv3 --kernel-trace --stats -- python tools/instance_rates.py   (200-frame streams, 4 launches per shape)" > $O
  stats 900 python $R/tools/instance_rates.py
  python $R/tools/rocprof_summary.py /tmp/pr | grep -v "at::\|rocclr" >> $O
  echo "# tool output under the profiler:" >> $O; grep -v amdgpu.ids /tmp/pr.log >> $O
  O=$R/gpurun_out/${tag}_instances_pmc.txt
  echo "# PMC passes (each its own run, --kernel-trace only) over tools/instance_rates.py; per kernel instance, means per shader engine (x32 for the chip) except FETCH/WRITE_SIZE (KiB, chip)" > $O
  for set in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    pmc 900 "$set" python $R/tools/instance_rates.py
    python $R/tools/pmc_extract.py /tmp/pm "fsk_demod_wave" | cut -c1-140 >> $O
  done
  unset PIRIP_RATES_WAVE_ONLY
fi
if has decim; then
  O=$R/gpurun_out/${tag}_decim.txt
  echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_decim.py --warmup 100 --iters 100" > $O
  stats 300 python $R/tools/bench_decim.py --warmup 100 --iters 100
  python $R/tools/rocprof_summary.py /tmp/pr >> $O 2>&1
  echo "# tool output under the profiler:" >> $O; grep stage /tmp/pr.log >> $O
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    pmc 300 "$set" python $R/tools/bench_decim.py --warmup 2 --iters 3
    python $R/tools/pmc_extract.py /tmp/pm decim >> $O
  done
  echo "# not profiled:" >> $O
  python $R/tools/bench_decim.py >> $O
fi
