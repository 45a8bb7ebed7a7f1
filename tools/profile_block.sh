#!/bin/bash
# tools/profile_block.sh TAG -- PMC passes over the Ts = 240 block instance (fsk_demod_block.hip) through tools/instance_rates.py:
#   gpurun -- 'bash tools/profile_block.sh r04_g'
# each counter set its own run with --kernel-trace only (MI355X_MICROARCH.md's recipe); values are means per shader engine
tag=${1:-r04_x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export PIRIP_RATES_BLOCK_STREAMS=6144 PIRIP_RATES_GENERAL_ONLY=1 PIRIP_RATES_BLOCK_ONLY=1
SETS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT")
O=$R/gpurun_out/${tag}_block_pmc.txt
echo "# PMC passes over: PIRIP_RATES_BLOCK_STREAMS=6144 python tools/instance_rates.py (block instance only: 6144 streams x 24 frames of 12000 samples, 2-FSK peak and 4-FSK mask); means per shader engine" > $O
for set in "${SETS[@]}"; do
  rm -rf /tmp/pm; timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/instance_rates.py > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm "block" | cut -c1-102 >> $O
done
tail -3 /tmp/pm.log >> $O
