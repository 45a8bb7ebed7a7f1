#!/bin/bash
# tools/chain_traffic.sh TAG -- HBM traffic of the config-4 whole chain (pirip_hip_fsk_ldpc_rx_batch, fused hand-over) per IQ sample:
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md's HBM recipe; FETCH_SIZE x2 on gfx950), summed over
# every kernel of one chain call, against the algorithmic 2 + 100/1200 bytes. PIRIP_CHAIN_ONLY makes tools/bench_configs.py run just
# the chain at the given Eb/N0.
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_chain_traffic.txt
echo "# config-4 chain (8192 streams x 600 000 samples, 4-FSK + FSK_LDPC, fused hand-over), per-kernel HBM bytes per chain call" > $O
for ebno in 7.0 3.5; do
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; PIRIP_CHAIN_ONLY=$ebno timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -- python $R/tools/bench_configs.py --iters 2 > /tmp/pm.log 2>&1
    echo "# Eb/N0 $ebno dB, $set (KiB per dispatch, mean over dispatches; calls = dispatches in the run: 1 warm-up + 2 timed chain calls)" >> $O
    python $R/tools/pmc_extract.py /tmp/pm "" | grep -v "^#\|^kernel\|at::\|copyBuffer" | cut -c1-48,52-110 >> $O
  done
done
python3 - "$O" <<'PY' >> $O
import sys, re
rows = {}
cur = None
for ln in open(sys.argv[1]):
    m = re.match(r"# Eb/N0 ([0-9.]+) dB, (\w+)", ln)
    if m: cur = (m.group(1), m.group(2)); rows.setdefault(cur, 0.0); continue
    if cur and not ln.startswith("#"):
        f = ln.split()
        try:
            n, mean = int(f[-2]), float(f[-1])
        except Exception:
            continue
        # every kernel of the chain runs once per call; the fill kernel (hipMemsetAsync of the payload and hard-decision words,
        # plus the handles' resets between calls) runs several times: its dispatches per call = n / 3 calls
        rows[cur] += mean * (n / 3.0) * 1024.0 * (2.0 if cur[1] == "FETCH_SIZE" else 1.0)
samples = 8192 * 600000 - 8192 * 0          # consumed samples differ by < 0.2 %
print("# ---- bytes per IQ sample (sum over the chain's kernels; FETCH_SIZE x 2: gfx950 correction) ----")
for eb in ("7.0", "3.5"):
    r, w = rows.get((eb, "FETCH_SIZE"), 0.0), rows.get((eb, "WRITE_SIZE"), 0.0)
    print(f"# Eb/N0 {eb} dB: read {r / samples:.4f} + written {w / samples:.4f} = {(r + w) / samples:.4f} B/sample; algorithmic 2.0833; ratio {(r + w) / samples / 2.083333:.3f}")
PY
cat $O | tail -40
