#!/bin/bash
# tools/batch_sweep.sh -- SURVEY.md 8d's batch-size sweep for config 2: B independent 2-FSK streams x 1.2 M samples resident in HBM,
# 20 timed launches after 3 warm-ups (bench.py's own clock: barrier + synchronize around the 20 steps; kernel_ms: HIP events per launch).
# One line per B. Below 3072 streams the chip is not full (one stream = one wavefront, 3072 resident), which is what the table shows.
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "# streams  G samples/s  ms per launch (HIP events)  fraction of HBM roofline (2.0058 B/sample)"
for B in 64 256 1024 3072 4096 6144 16384; do
  python $R/bench.py --streams $B --steps 20 --warmup 3 --no-cpu-baseline --no-extra --check-streams 8 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%8d  %10.1f  %10.3f  %8.4f  bit_errors_vs_cpu_ref %d' % ($B, j['value'] / 1e3, j['roofline']['kernel_ms'], j['roofline']['frac'], j['bit_errors_vs_cpu_ref']))"
done
