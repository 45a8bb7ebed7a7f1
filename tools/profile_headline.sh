#!/bin/bash
# tools/profile_headline.sh -- the rocprofv3 passes behind profiles/r01_*_fast_kernel_*.txt (run on the GPU box:
#   gpurun -- 'bash tools/profile_headline.sh r01_j'); writes text summaries under gpurun_out/, raw .db files stay in /tmp.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
O=$R/gpurun_out/${tag}_fast_kernel_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline" > $O
rm -rf /tmp/pr; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr >> $O 2>&1
echo "# bench line under the profiler:" >> $O; grep '^{' /tmp/pr.log | cut -c1-900 >> $O
O=$R/gpurun_out/${tag}_fast_kernel_pmc.txt
# (counter passes at 4096 streams: rocprofv3 --pmc segfaults on the 16384-stream default dispatch -- 1 M workgroups;
#  the counters are used per sample / per frame, which do not depend on the stream count)
echo "# PMC passes, each its own run of: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --streams ${PMC_STREAMS:-4096} --steps 2 --warmup 1 --no-cpu-baseline" > $O
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/pm; timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/bench.py --streams ${PMC_STREAMS:-4096} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm fsk_demod >> $O
done
