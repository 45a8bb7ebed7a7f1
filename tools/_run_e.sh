cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/e_chain_unsplit_stats.txt; mkdir -p $R/gpurun_out/r06
echo "# chain_ab child at 3.5 dB, PIRIP_CHAIN_SPLIT_MIN=0, kernel trace per decoder" > $O
for dec in fast bank; do
  echo "## PIRIP_LDPC_DECODER=$dec" >> $O
  rm -rf /tmp/pr; PIRIP_CHAIN_SPLIT_MIN=0 PIRIP_LDPC_DECODER=$dec AB_ITERS=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/chain_ab.py --child 3.5 > /tmp/pr.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/pr | head -14 >> $O
  grep ABCHAIN /tmp/pr.log >> $O
done
cat $O
