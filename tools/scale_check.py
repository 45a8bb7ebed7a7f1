#!/usr/bin/env python3
"""Decision-level comparison of the device with the CPU oracle at scale: B individually noised streams made by
the device-side Tx, one pass of the batch demodulator from the reset state, every stream replayed by the oracle on the host
cores, every differing bit classified.

  inside  -- the ORACLE's own decision margin (largest minus second largest tone magnitude of that symbol) is below
             NEAR_TIE of the stream's peak magnitude: the two float32 evaluation orders straddle a tie
  illcond -- not a near-tie, but in a frame whose fine-timing ESTIMATES differ by more than TIMING_TIE symbols between the two:
             the estimate is the angle of a sum of (Nsym+1)*P terms that nearly cancels when the symbol-rate line is weak (a
             few frames in 10^4 at 3-5 dB), so summation-order differences of 1e-6 relative move the angle by 1e-4..1e-2
             symbols and with it every interpolated magnitude of the frame (the largest rx_filt errors of a run are these
             frames); a decision with a small -- not tiny -- margin can then differ. Reported, bounded, not hidden.
  outside -- any other differing bit (none may exist)
  first   -- of all differing bits, those in the very first decision of a stream (frame 0, symbol 0): a recording that
             starts mid-symbol hands that decision a fraction of a symbol
  timing  -- a stream whose nin SEQUENCE parts from the oracle's: the fine-timing estimate of one frame sat on a decision
             threshold (+-0.25 symbols, where nin changes by Ts/4) closer than TIMING_TIE, the two float32 evaluation orders
             fall on different sides of it, and from the next frame on the two demodulators look at different sample
             windows (they re-converge a few frames later, but nothing after the split is comparable bit for bit). Bits are
             compared up to and including the frame of the split; the split itself must be such a near-tie, anything else
             counts as `unexplained`.

Used by tests/test_scale_check.py (-m gpu) with committed bounds, and stand-alone to write the profiles/ table:
  python tools/scale_check.py [--streams 2048] [--samples 1200000] [--detail] > profiles/r04_scale_check.txt
The oracle is the checker here; nothing in the product imports this file."""
import argparse
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FS, RS, NSYM = 240000, 10000, 50
TS = FS // RS
F1, SHIFT = 10000, 10000
N_PLANS = 5
NEAR_TIE = 2e-4
TIMING_TIE = 5e-5      # symbols; tests/test_gpu_parity.py TIMING_TOL

_G = None


def _worker(k):
    """oracle replay of stream k and classification of the device's differing bits"""
    from oracle import binding as ob
    g = _G
    M = g["M"]
    rx = ob.OracleFsk(FS, g["rs"], M, P=g["P"], est_min=g["est_min"], est_max=g["est_max"], tone_spacing=g["mask"] if g["mask"] else 100, mask=bool(g["mask"]))
    ro = rx.demod(g["iq"][k], ob.IN_CU8_FSKDEMOD, want_filt=True, want_stats=True)
    n = ro["nframes"]
    hb, hf, hs = g["bits"][k], g["filt"][k], g["stats"][k]
    res = {"nframes": n, "dev_nframes": int(g["nfr"][k]), "inside": 0, "outside": 0, "illcond": 0, "first": 0, "detail": [],
           "nin_equal": True, "fest_equal": True, "filt_err": 0.0, "bits": n * hb.shape[1]}
    if int(g["nfr"][k]) != n:
        res["outside"] = -1
        return res
    f = ro["rx_filt"].reshape(n, M, NSYM)
    peak = float(np.abs(f).max()) if n else 1.0
    res["nin_equal"] = bool(np.array_equal(ro["stats"][:, 6], hs[:n, 6]))
    res["split"] = None
    if not res["nin_equal"]:
        f0 = int(np.nonzero(ro["stats"][:, 6] != hs[:n, 6])[0][0])          # first frame whose nin_next differs
        t_o, t_d = float(ro["stats"][f0, 4]), float(hs[f0, 4])
        dist = min(abs(abs(t_o) - 0.25), abs(abs(t_d) - 0.25), abs(abs(t_o) - 0.5), abs(abs(t_d) - 0.5))   # +-0.25: nin changes; +-0.5: atan2's wrap
        res["split"] = (int(k), f0, t_o, t_d, dist, n - 1 - f0)
        # the split frame itself saw the same samples on both sides and is compared -- unless the split is atan2's wrap at +-0.5,
        # where the two estimates are one whole symbol apart and that frame's decisions already sample different symbols
        wrap = min(abs(abs(t_o) - 0.5), abs(abs(t_d) - 0.5)) < TIMING_TIE
        n = f0 if wrap else f0 + 1
        f = f[:n]
    res["fest_equal"] = bool(np.array_equal(ro["stats"][:n, :4], hs[:n, :4]))
    err = np.abs(hf[:n].reshape(n, M, NSYM) - f) / peak if n else np.zeros((0, M, NSYM))
    res["filt_err"] = float(err.max()) if n else 0.0
    fmax = err.reshape(n, -1).max(axis=1) if n else np.zeros(0)
    res["timing_illcond_frames"] = int((np.abs(ro["stats"][:n, 4] - hs[:n, 4]) > TIMING_TIE).sum())
    res["frames"] = int(n); res["frames_over_1e4"] = int((fmax > 1e-4).sum()); res["frames_over_1e3"] = int((fmax > 1e-3).sum())
    probe = g.get("probe")
    if probe and probe[0] == k:
        res["probe"] = [(fr, float(ro["stats"][fr, 4]), float(hs[fr, 4]), [float(v) for v in ro["stats"][fr, :M]], [float(v) for v in hs[fr, :M]],
                         float(ro["stats"][fr, 6]), float(fmax[fr]), [float(v) for v in err[fr].max(axis=0)[:6]], [float(v) for v in err[fr].max(axis=0)[-3:]],
                         float(ro["stats"][fr, 5]), float(hs[fr, 5])) for fr in range(max(probe[1] - 4, 0), min(probe[1] + 3, n))]
    if n and res["filt_err"] > 2e-4:
        fr, mm, sy = np.unravel_index(int(err.argmax()), err.shape)
        res["worst"] = (int(k), int(fr), int(mm), int(sy), res["filt_err"], float(f[fr, mm, sy]) / peak, float(hf[fr].reshape(M, NSYM)[mm, sy]) / peak,
                        float(ro["stats"][fr, 4]), float(hs[fr, 4]), [float(v) for v in ro["stats"][max(fr - 1, 0):fr + 1, 6]],
                        float(err[fr].max()), float(np.median(err[fr])), [float(v) for v in ro["stats"][fr, :M]],
                        [float(v) for v in ro["stats"][max(fr - 1, 0), :M]])
    res["bits"] = n * hb.shape[1]
    diff = np.argwhere(hb[:n] != ro["bits"][:n])
    bps = 1 if M == 2 else 2
    seen = set()
    for fr, b in diff:
        sym = int(b) // bps
        srt = np.sort(f[fr, :, sym])
        margin = float(srt[-1] - srt[-2]) / peak
        inside = margin < NEAR_TIE
        illcond = abs(float(ro["stats"][fr, 4]) - float(hs[fr, 4])) > TIMING_TIE
        res["inside" if inside else "illcond" if illcond else "outside"] += 1
        if fr == 0 and sym == 0:
            res["first"] += 1
        if (fr, sym) not in seen and len(res["detail"]) < 8:
            seen.add((fr, sym))
            hfs = hf[fr].reshape(M, NSYM)[:, sym]
            res["detail"].append((int(k), int(fr), sym, margin, [float(v) / peak for v in f[fr, :, sym]],
                                  [float(v) / peak for v in hfs], float(ro["stats"][fr, 4]), float(hs[fr, 4])))
    return res


def _usable_cores():
    """workers worth starting: the affinity mask, cut to the cgroup's CPU quota (a GPU box shows 256 cores and grants ~16: 256 forked workers
    then time-slice and the replay takes 40 s instead of 10)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def run(M=2, P=24, ebno_db=None, nstreams=2048, nsamp=1_200_000, seed=0x5eed, est=None, procs=None, probe=None, rs=RS, mask=0):
    """Returns the classification summed over all streams (dict). rs: symbol rate (10000: Ts = 24, the wave instances; 1000: Ts = 240,
    the block instance of `rtl_fsk -r 1000`, tones 2 kHz apart as README.md:239's --mask 2000 implies); mask: the mask estimator's spacing."""
    import torch
    import pirip_amd
    from pirip_amd.binding import synth_cu8, STATS_PER_FRAME
    est_min, est_max = est if est else ((500, 25000) if M == 2 else (500, 60000))
    bps = 1 if M == 2 else 2
    ts = FS // rs
    f1, shift = (F1, SHIFT) if rs == RS else (11000, 2000)
    nsym = (nsamp + ts) // ts + NSYM
    nsym -= nsym % NSYM
    bin_path = os.path.join(ROOT, "pirip_amd", "bin", "fsk_get_test_bits")
    txbits = np.frombuffer(subprocess.run([bin_path, "-", str(nsym * bps)], capture_output=True, check=True).stdout,
                           dtype=np.uint8)[:nsym * bps].copy()
    B = nstreams
    gsi = np.arange(B)
    f1s = f1 + np.rint(((gsi % N_PLANS) - 2) * 937.5).astype(np.int32)
    skips = ((gsi // N_PLANS) % ts).astype(np.int32)
    sigma = 0.0 if ebno_db is None else float(np.sqrt((4.0 * ts / bps / 10 ** (ebno_db / 10.0)) / 2.0))
    amp = 32.0 if ebno_db is None else (16.0 if ebno_db >= 10 else 8.0)
    if ebno_db is not None and ts > TS:
        amp *= float(np.sqrt(TS / ts))            # same noise level in LSBs as at Ts = 24: the u8 range does not clip it
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    dtx = torch.from_numpy(txbits).cuda()
    synth_cu8(FS, rs, M, f1s, shift, dtx.data_ptr(), 0, nsym, dev.data_ptr(), nsamp * 2, nsamp,
              amp=amp, sigma=sigma, seed=seed, skip=skips, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = pirip_amd.HipDemod(FS, rs, M, P=P, Nsym=NSYM, est_min=est_min, est_max=est_max, mask=mask,
                           in_format=pirip_amd.IN_CU8_FSKDEMOD, nstreams=B)
    maxf = h.max_frames_for(nsamp)
    bits = torch.zeros((B, maxf, h.Nbits), dtype=torch.uint8, device="cuda")
    filt = torch.zeros((B, maxf, M * NSYM), dtype=torch.float32, device="cuda")
    stats = torch.zeros((B, maxf, STATS_PER_FRAME), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * h.Nbits, filt.data_ptr(), maxf * M * NSYM,
                  stats.data_ptr(), maxf * STATS_PER_FRAME, nfr.data_ptr(), cons.data_ptr(), maxf,
                  torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    global _G
    _G = {"M": M, "P": P, "rs": rs, "mask": mask, "est_min": est_min, "est_max": est_max, "iq": dev.cpu().numpy(), "bits": bits.cpu().numpy(),
          "filt": filt.cpu().numpy(), "stats": stats.cpu().numpy(), "nfr": nfr.cpu().numpy(), "probe": probe}
    kernel = h.kernel_name()
    del dev, bits, filt, stats, h
    torch.cuda.empty_cache()
    ncore = procs or _usable_cores()
    with mp.get_context("fork").Pool(min(ncore, B)) as pool:
        reps = pool.map(_worker, range(B), chunksize=max(1, B // (4 * ncore)))
    out = {"M": M, "P": P, "rs": rs, "mask": mask, "ebno_db": ebno_db, "streams": B, "samples": nsamp, "kernel": kernel,
           "bits": sum(r["bits"] for r in reps), "inside": sum(r["inside"] for r in reps),
           "outside": sum(max(r["outside"], 0) for r in reps), "illcond": sum(r["illcond"] for r in reps), "first": sum(r["first"] for r in reps),
           "timing_illcond_frames": sum(r.get("timing_illcond_frames", 0) for r in reps),
           "frame_count_mismatch": sum(1 for r in reps if r["outside"] < 0),
           "nin_mismatch_streams": sum(1 for r in reps if not r["nin_equal"]),
           "timing_splits": [r["split"] for r in reps if r.get("split")],
           "unexplained_splits": sum(1 for r in reps if r.get("split") and not r["split"][4] < TIMING_TIE),
           "frames_after_splits": sum(r["split"][5] for r in reps if r.get("split")),
           "fest_mismatch_streams": sum(1 for r in reps if not r["fest_equal"]),
           "max_filt_err": max(r["filt_err"] for r in reps),
           "first_diffs_on_zero_offset_streams": sum(r["first"] for k, r in enumerate(reps) if skips[k] == 0),
           "frames_compared": sum(r.get("frames", 0) for r in reps), "frames_over_1e4": sum(r.get("frames_over_1e4", 0) for r in reps),
           "frames_over_1e3": sum(r.get("frames_over_1e3", 0) for r in reps), "probe": [r["probe"] for r in reps if r.get("probe")],
           "worst": sorted([r["worst"] for r in reps if r.get("worst")], key=lambda w: -w[4])[:12],
           "detail": [d for r in reps for d in r["detail"]]}
    _G = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=1_200_000)
    ap.add_argument("--detail", action="store_true")
    ap.add_argument("--cases", default="2:24:none,2:24:6,2:24:3,4:8:none,4:8:7,4:8:5")
    ap.add_argument("--json", default=None, help="also write the per-case results (without the detail lists) to this file")
    ap.add_argument("--probe", default=None, help="stream:frame -- print that stream's frames around it (timing, f_est, per-symbol rx_filt error)")
    a = ap.parse_args()
    print("# tools/scale_check.py: device vs oracle, one pass from the reset state, every stream replayed on the host")
    print("# M P Eb/N0 streams bits compared | differing bits: inside the near-tie rule, in frames with an ill-conditioned timing estimate, outside both (must be 0), of all those in a stream's first decision "
          "(... on streams with start offset 0) | streams whose nin sequence splits at a timing near-tie (unexplained splits; frames after "
          "the splits, not compared) / streams whose f_est differ | max rx_filt error (of the stream's peak; frames whose largest error exceeds 1e-4 / 1e-3) | kernel")
    allr = {}
    for c in a.cases.split(","):
        m, p, e = c.split(":")[:3]
        cs = c.split(":")                          # M:P:Eb/N0[:streams[:Rs[:mask spacing[:samples]]]]
        nstr = int(cs[3]) if len(cs) > 3 else a.streams
        rs = int(cs[4]) if len(cs) > 4 else RS
        mask = int(cs[5]) if len(cs) > 5 else 0
        nsmp = int(cs[6]) if len(cs) > 6 else a.samples
        r = run(int(m), int(p), None if e == "none" else float(e), nstr, nsmp, probe=tuple(int(v) for v in a.probe.split(":")) if a.probe else None, rs=rs, mask=mask)
        print(f"{r['M']} {r['P']} {e} {r['streams']} {r['bits']} | {r['inside']} {r['illcond']} {r['outside']} {r['first']} "
              f"({r['first_diffs_on_zero_offset_streams']}) | {r['nin_mismatch_streams']} ({r['unexplained_splits']}; {r['frames_after_splits']}) / {r['fest_mismatch_streams']} "
              f"| {r['max_filt_err']:.2e} (frames over 1e-4: {r['frames_over_1e4']}, over 1e-3: {r['frames_over_1e3']}, frames whose timing estimates differ by more than {TIMING_TIE:g}: {r['timing_illcond_frames']}, of {r['frames_compared']}) | {r['kernel']}", flush=True)
        for pr in r["probe"]:
            for q in pr:
                print(f"#   probe frame {q[0]} timing {q[1]:+.6f} / {q[2]:+.6f} f_est {q[3]} / {q[4]} nin_next {q[5]:.0f} frame max err {q[6]:.2e} first syms {['%.1e' % v for v in q[7]]} "
                      f"last {['%.1e' % v for v in q[8]]} SNRest {q[9]:.4f} / {q[10]:.4f}")
        allr[c if len(cs) > 4 else f"{m}:{p}:{e}"] = {k: v for k, v in r.items() if k not in ("detail", "worst", "probe")}
        for sp in r["timing_splits"]:
            wrap = min(abs(abs(sp[2]) - 0.5), abs(abs(sp[3]) - 0.5)) < min(abs(abs(sp[2]) - 0.25), abs(abs(sp[3]) - 0.25))
            print(f"#   split: stream {sp[0]} frame {sp[1]} norm_rx_timing oracle {sp[2]:+.7f} device {sp[3]:+.7f} "
                  f"(distance to {'the atan2 wrap at +-0.5' if wrap else 'the nin threshold +-0.25'}: {sp[4]:.1e}), {sp[5]} later frames not compared")
        if a.detail:
            for w in r["worst"]:
                print(f"#   worst rx_filt: stream {w[0]} frame {w[1]} tone {w[2]} sym {w[3]} err {w[4]:.2e} oracle {w[5]:.5f} device {w[6]:.5f} timing {w[7]:+.6f} / {w[8]:+.6f} "
                      f"nin_next(prev, this) {w[9]} frame max / median err {w[10]:.2e} / {w[11]:.2e} f_est {w[12]} prev {w[13]}")
            for d in r["detail"][:60]:
                print(f"#   stream {d[0]} frame {d[1]} sym {d[2]} margin {d[3]:.2e} oracle {['%.5f' % v for v in d[4]]} "
                      f"device {['%.5f' % v for v in d[5]]} timing {d[6]:+.6f} / {d[7]:+.6f}")
    if a.json:
        import json
        json.dump(allr, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
