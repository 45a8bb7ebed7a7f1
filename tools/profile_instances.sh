#!/bin/bash
# tools/profile_instances.sh TAG -- rocprofv3 kernel trace + PMC passes over every wave-kernel instance family
# (tools/instance_rates.py, wave kernel only): one row per kernel template instance.   gpurun -- 'bash tools/profile_instances.sh r02_e'
tag=${1:-r02_x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export PIRIP_RATES_WAVE_ONLY=1
O=$R/gpurun_out/${tag}_instances_stats.txt
echo "# PIRIP_RATES_WAVE_ONLY=1 rocprofv3 --kernel-trace --stats -- python tools/instance_rates.py   (200-frame streams, 4 launches per shape)" > $O
rm -rf /tmp/pr; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/instance_rates.py > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr | grep -v "at::\|rocclr" >> $O
echo "# tool output under the profiler:" >> $O; grep -v amdgpu.ids /tmp/pr.log >> $O
O=$R/gpurun_out/${tag}_instances_pmc.txt
echo "# PMC passes (each its own run, --kernel-trace only) over tools/instance_rates.py; per kernel instance, means per shader engine (x32 for the chip) except FETCH/WRITE_SIZE (KiB, chip)" > $O
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/pm; timeout 900 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/instance_rates.py > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm "fsk_demod_wave" | cut -c1-140 >> $O
done
