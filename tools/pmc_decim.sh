cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01_i_decim.txt
echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_decim.py --warmup 100 --iters 100" > $O
rm -rf /tmp/pr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/bench_decim.py --warmup 100 --iters 100 > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr >> $O 2>&1
echo "# tool output under the profiler:" >> $O; cat /tmp/pr.log | grep stage >> $O
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/bench_decim.py --warmup 2 --iters 3 > /tmp/pm.log 2>&1
  python $R/tools/pmc_extract.py /tmp/pm decim >> $O
done
echo "# not profiled:" >> $O
python $R/tools/bench_decim.py >> $O
