#!/bin/bash
# tools/batch_sweep_tail.sh -- config 2 at stream counts around multiples of the 3072 resident waves (12288 ... 30720): is there a partial
# last round of workgroups worth filling? (profiles/r05_s_batch_sweep_tail.txt: no, the rate rises monotonically)
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "# streams  G samples/s  ms per launch (HIP events)  fraction of HBM roofline"
for B in 12288 15360 16384 18432 21504 24576 30720; do
  python $R/bench.py --streams $B --steps 10 --warmup 2 --no-cpu-baseline --no-extra --check-streams 8 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%8d  %10.1f  %10.3f  %8.4f  bit_errors_vs_cpu_ref %d' % ($B, j['value'] / 1e3, j['roofline']['kernel_ms'], j['roofline']['frac'], j['bit_errors_vs_cpu_ref']))"
done
