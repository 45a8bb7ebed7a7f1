cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/bench.py --streams ${PMC_STREAMS:-4096} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm fsk_demod | cut -c52-110
done
