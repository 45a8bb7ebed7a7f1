cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01_l_configs_3_4_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py   (config 4 demod half: 8192 x 600k samples; config 3: 4096 streams x 1.8e6 u8 samples -> /45 -> demod)" > $O
rm -rf /tmp/pr; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr -- python $R/tools/bench_configs.py --iters 20 > /tmp/pr.log 2>&1
python $R/tools/rocprof_summary.py /tmp/pr 2>&1 | head -8 >> $O
echo "# tool output under the profiler:" >> $O; grep '^{' /tmp/pr.log >> $O
