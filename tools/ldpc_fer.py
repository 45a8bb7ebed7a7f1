#!/usr/bin/env python3
"""What the product's FSK_LDPC precision choices cost (/root/reference/README.md:200-212,241-251).

B streams of continuous coded frames (the repo's framer, stand-in (512,256) code) are modulated and individually noised on the
device, demodulated once with soft magnitudes out, and received FOUR ways from those same magnitudes / the same IQ:
  gpu      the product: pirip_hip_fsk_ldpc_rx_batch, IQ -> records in one call (binary16 soft bits, wave-order sums, table phi)
  mirror   oracle/ldpc_oracle.c: the CPU statement of exactly that arithmetic (must equal gpu record for record)
  indep    oracle/ldpc_independent.c mode 1: float32 soft bits, serial sums, exact ln I0, double-precision sum-product
  recalled oracle/ldpc_independent.c mode 2: codec2's fsk_rx_filt_to_llrs as recalled [UPSTREAM-RECALLED], same decoder as indep
and every receiver is scored against the TRANSMITTED payloads: frame error rate (frames not delivered with a good CRC and
the right bytes), undetected errors (CRC good, bytes wrong), bit error rate of all decoded payloads, mean iterations.

  python tools/ldpc_fer.py [--streams 1088] [--frames 104] [--ebno 3.5,5,7] > profiles/r04_ldpc_fer.txt
Eb/N0 is per CHANNEL bit (Es / log2 M), the convention of tools/bench_configs.py and tests/test_ldpc.py. Checker only."""
import argparse
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RX_BITS = 4
FS, RS, TS, NSYM = 240000, 10000, 24, 50
_G = None


def _score(status, payload, info, expected, lo, hi):
    """-> (delivered[frames] bool, undetected, bit errors, bits compared, iteration sum, decodes)"""
    nfr = expected.shape[0]
    delivered = np.zeros(nfr, dtype=bool)
    undetected = berr = bcnt = itsum = ndec = 0
    exp_bits = np.unpackbits(expected, axis=1)
    for c in np.nonzero(info[:, 6] >= 0)[0]:
        pb = np.unpackbits(payload[c])
        dist = (exp_bits[:, 16:240] != pb[16:240]).sum(axis=1)          # bytes 2..29: the data common to all frames but seq/CRC
        seq = int(payload[c, 1]) - 1
        f = seq if (0 <= seq < nfr and dist[seq] <= 60) else int(np.argmin((exp_bits != pb).sum(axis=1)))
        d = int((exp_bits[f] != pb).sum())
        if d > 100 or not (lo <= f < hi):
            continue                                                    # a false lock in the noise / a frame outside the scored range
        ndec += 1; berr += d; bcnt += 256; itsum += int(info[c, 4])
        if status[c] & RX_BITS:
            if d == 0:
                delivered[f] = True
            else:
                undetected += 1
    return delivered[lo:hi], undetected, berr, bcnt, itsum, ndec


def _worker(s):
    from oracle import binding as ob
    g = _G
    code, M = g["code"], g["M"]
    filt = g["filt"][s, :g["nfr"][s]]
    out = {}
    rx = {"mirror": ob.OracleLdpc(code, M), "indep": ob.IndepLdpc(code, M, mode=1), "recalled": ob.IndepLdpc(code, M, mode=2),
          "recalled_phi0": ob.IndepLdpc(code, M, mode=3)}
    recs = {k: r.rx(filt) for k, r in rx.items()}
    n = g["nfr"][s]
    recs["gpu"] = (g["st"][s, :n], g["pl"][s, :n], g["inf"][s, :n])
    for k, (st, pl, inf) in recs.items():
        out[k] = _score(st, pl, inf, g["expected"], g["lo"], g["hi"])
    out["gpu_equals_mirror"] = bool(np.array_equal(recs["gpu"][0], recs["mirror"][0]) and np.array_equal(recs["gpu"][1], recs["mirror"][1])
                                    and np.array_equal(recs["gpu"][2][:, :9], recs["mirror"][2][:, :9]))
    return out


def run(ebno_db, streams=1024, frames=104, M=4, P=8, seed=0xfec, procs=None, llr_map=None):
    import torch
    import pirip_amd
    from pirip_amd.binding import synth_cu8
    from oracle import binding as ob
    code_path = pirip_amd.STANDIN_CODE
    if llr_map:                                  # the code file's llr_map key (default when absent: upstream)
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sigutil
        code_path = sigutil.code_variant(pirip_amd.STANDIN_CODE, tempfile.mkdtemp(), llr_map)
    code = ob.parse_code_file(code_path)
    bps = 1 if M == 2 else 2
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = np.frombuffer(subprocess.run([framer, "--code", pirip_amd.STANDIN_CODE, "-m", str(M), "--testframes", str(frames), "--bursts", "1",
                                       "--seq", "--source", "0x1", "/dev/zero", "-"], capture_output=True, check=True).stdout, dtype=np.uint8)
    pre = 50 * bps * bps                     # preamble_bits(): 50 * (M/2) symbols (rpitx_fsk.cpp:313-336 sends npreamble*bps symbols' worth)
    bpf = 32 + code["n"]
    assert fb.size == pre + frames * bpf, (fb.size, pre, frames, bpf)
    expected = np.stack([np.packbits(fb[pre + f * bpf + 32:][:code["k"]]) for f in range(frames)])
    nsym = fb.size // bps
    B = streams
    nsamp = nsym * TS - TS
    gsi = np.arange(B)
    f1s = 10000 + np.rint(((gsi % 5) - 2) * 937.5).astype(np.int32)
    skips = ((gsi // 5) % TS).astype(np.int32)
    sigma = float(np.sqrt((4.0 * TS / bps) / (10 ** (ebno_db / 10.0)) / 2.0))
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    dtx = torch.from_numpy(fb.copy()).cuda()
    synth_cu8(FS, RS, M, f1s, 10000, dtx.data_ptr(), 0, nsym, dev.data_ptr(), nsamp * 2, nsamp, amp=14.0, sigma=sigma, seed=seed,
              skip=skips, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    est_max = 25000 if M == 2 else 60000
    h = pirip_amd.HipDemod(FS, RS, M, P=P, est_min=500, est_max=est_max, nstreams=B)
    maxf = h.max_frames_for(nsamp)
    per = M * NSYM
    filt = torch.zeros((B, maxf, per), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    cs = torch.cuda.current_stream().cuda_stream
    h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, 0, 0, filt.data_ptr(), maxf * per, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, cs)
    torch.cuda.synchronize()
    h.reset(cs)
    ld = pirip_amd.HipLdpc(code_path, M, nstreams=B)
    st = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda")
    pl = torch.zeros((B, maxf, code["k"] // 8), dtype=torch.uint8, device="cuda")
    inf = torch.zeros((B, maxf, pirip_amd.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")
    nfr2 = torch.zeros(B, dtype=torch.int32, device="cuda")
    ld.chain_batch(h, dev.data_ptr(), nsamp * 2, nsamp, st.data_ptr(), pl.data_ptr(), inf.data_ptr(), nfr2.data_ptr(), cons.data_ptr(), maxf, stream=cs)
    torch.cuda.synchronize()
    fused = ld.last_path_fused()
    assert torch.equal(nfr, nfr2)
    global _G
    _G = {"code": code, "M": M, "filt": filt.cpu().numpy(), "nfr": nfr.cpu().numpy(), "st": st.cpu().numpy(), "pl": pl.cpu().numpy(),
          "inf": inf.cpu().numpy(), "expected": expected, "lo": 4, "hi": frames - 4}
    del dev, filt, st, pl, inf, h, ld
    torch.cuda.empty_cache()
    import scale_check
    ncore = procs or scale_check._usable_cores()          # affinity mask cut to the cgroup CPU quota
    with mp.get_context("fork").Pool(min(ncore, B)) as pool:
        reps = pool.map(_worker, range(B), chunksize=max(1, B // (8 * ncore)))
    _G = None
    scored = B * (frames - 8)
    res = {"ebno_db": ebno_db, "M": M, "P": P, "streams": B, "frames_scored": scored, "fused": fused,
           "gpu_equals_mirror_streams": sum(r["gpu_equals_mirror"] for r in reps), "llr_map": code["llr_map"]}
    dl = {}
    for k in ("gpu", "mirror", "indep", "recalled", "recalled_phi0"):
        d = np.stack([r[k][0] for r in reps])
        dl[k] = d
        und = sum(r[k][1] for r in reps); be = sum(r[k][2] for r in reps); bc = sum(r[k][3] for r in reps)
        its = sum(r[k][4] for r in reps); nd = sum(r[k][5] for r in reps)
        res[k] = {"fer": 1.0 - float(d.mean()), "frame_errors": int((~d).sum()), "undetected": und, "ber_decoded": be / max(bc, 1),
                  "bit_errors": be, "decodes": nd, "mean_iter": its / max(nd, 1)}
    res["gpu_only_vs_indep"] = int((dl["gpu"] & ~dl["indep"]).sum())      # frames the product delivered and the independent receiver lost
    res["indep_only_vs_gpu"] = int((~dl["gpu"] & dl["indep"]).sum())
    res["recalled_only_vs_gpu"] = int((~dl["gpu"] & dl["recalled"]).sum())
    res["gpu_only_vs_recalled"] = int((dl["gpu"] & ~dl["recalled"]).sum())
    res["gpu_only_vs_recalled_phi0"] = int((dl["gpu"] & ~dl["recalled_phi0"]).sum())
    res["recalled_phi0_only_vs_gpu"] = int((~dl["gpu"] & dl["recalled_phi0"]).sum())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1088)
    ap.add_argument("--frames", type=int, default=104)
    ap.add_argument("--ebno", default="3.5,5,7")
    ap.add_argument("--M", type=int, default=4)
    ap.add_argument("--P", type=int, default=8)
    ap.add_argument("--llr-map", default=None, help="upstream | rician: the product's (and the mirror's) soft-decision mapping; default: the code file's (upstream)")
    a = ap.parse_args()
    print("# tools/ldpc_fer.py: FSK_LDPC receive, stand-in (512,256) code, %d-FSK Fs=240k Rs=10k P=%d, %d streams x %d frames "
          "(the first and last 4 of a stream are not scored: acquisition / tail)" % (a.M, a.P, a.streams, a.frames))
    print("# product llr_map: %s (upstream = codec2's fsk_rx_filt_to_llrs as recalled [UPSTREAM-RECALLED], the default since round 5; rician = exact ln I0, rounds 2-4)" % (a.llr_map or "code file default = upstream"))
    print("# receivers: gpu = product (its llr_map, binary16 soft bits, wave-order sums, table phi); mirror = CPU statement of the same arithmetic; "
          "indep = float32 soft bits, serial sums, exact ln I0, double sum-product; recalled = codec2's fsk_rx_filt_to_llrs as recalled "
          "[UPSTREAM-RECALLED] + the same double sum-product; recalled_phi0 = that with the decoder's phi limited to codec2's phi0() range as recalled (x < 9.08e-5 -> 10, x > 10 -> 0): what the product's decoder range follows since round 5")
    print("# Eb/N0(dB,channel bit) receiver frames_scored frame_errors FER undetected decoded_BER mean_iterations")
    for e in a.ebno.split(","):
        r = run(float(e), a.streams, a.frames, a.M, a.P, llr_map=a.llr_map)
        for k in ("gpu", "mirror", "indep", "recalled", "recalled_phi0"):
            v = r[k]
            print(f"{e} {k:13s} {r['frames_scored']} {v['frame_errors']} {v['fer']:.3e} {v['undetected']} {v['ber_decoded']:.3e} {v['mean_iter']:.2f}")
        print(f"# {e} dB: fused hand-over {r['fused']}; streams whose gpu records equal the mirror's: {r['gpu_equals_mirror_streams']} of {r['streams']}; "
              f"frames delivered by gpu only / indep only: {r['gpu_only_vs_indep']} / {r['indep_only_vs_gpu']}; gpu only / recalled only: "
              f"{r['gpu_only_vs_recalled']} / {r['recalled_only_vs_gpu']}; gpu only / recalled_phi0 only: {r['gpu_only_vs_recalled_phi0']} / {r['recalled_phi0_only_vs_gpu']}", flush=True)


if __name__ == "__main__":
    main()
