#!/bin/bash
# tools/pmc_decode.sh TAG [ebno] -- FSK_LDPC decoder kernels under rocprofv3 on the config-4 workload (tools/chain_ab.py's child: 8192 streams x
# 600k samples, whole chain + stand-alone receive stage), once per decoder choice (PIRIP_LDPC_DECODER=fast / bank): kernel-trace stats, then
# the LDS / VALU counter sets, each in its own run (--kernel-trace --pmc only).
tag=${1:-r06_x}; ebno=${2:-3.5}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_decode_pmc.txt; mkdir -p $(dirname $O)
echo "# tools/pmc_decode.sh: python tools/chain_ab.py --child $ebno (AB_ITERS=2), per decoder; counters are means per dispatch and shader engine" > $O
SETS=("GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC")
for dec in ${DECODERS:-fast bank}; do
  echo "## PIRIP_LDPC_DECODER=$dec $EXTRA_ENV" >> $O
  rm -rf /tmp/pr; env $EXTRA_ENV PIRIP_LDPC_DECODER=$dec AB_ITERS=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/chain_ab.py --child $ebno > /tmp/pr.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pr | head -12 >> $O
  grep ABCHAIN /tmp/pr.log >> $O
  for set in "${SETS[@]}"; do
    rm -rf /tmp/pm; env $EXTRA_ENV PIRIP_LDPC_DECODER=$dec AB_ITERS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/chain_ab.py --child $ebno > /tmp/pm.log 2>&1
    python $R/tools/pmc_extract.py /tmp/pm "decode_" | cut -c1-48,52-140 >> $O
  done
done
