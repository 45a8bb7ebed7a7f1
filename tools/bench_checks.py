#!/usr/bin/env python3
"""tools/bench_checks.py -- everything bench.py reports BESIDE the timed region: the CPU leg (`cpu_baseline`), the oracle replay
of the device's bits, the counter-derived annotations of the roofline object, and the side figures (opt-in band-only estimator,
the exact-order kernel, the other BASELINE configurations). Nothing in here runs between bench.py's two clock reads.

The oracle (oracle/) is imported here only as the CPU baseline leg and as the checker of the product's output -- never as a
compute path of the product."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NEAR_TIE = 2e-4      # of the stream's peak magnitude: the rule of tests/test_gpu_parity.py::_compare

_CPU_BUF = None      # the bench's sample buffer, inherited by the forked CPU-leg workers
_CHK = None          # (buffers, passes) of the forked replay workers
_CFG = None          # bench.py's (FS, RS, M, P, NSYM, EST_MIN, EST_MAX)


def configure(cfg, cpu_buf=None):
    global _CFG, _CPU_BUF
    _CFG = cfg
    if cpu_buf is not None:
        _CPU_BUF = cpu_buf


def cgroup_cpu_quota():
    """cores' worth of CPU time the container may use (cpu.max / cfs quota), or None"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                return q / per
        except Exception:
            pass
    return None


def _oracle_rx():
    from oracle import binding as ob
    FS, RS, M, P, NSYM, EST_MIN, EST_MAX = _CFG
    return ob, ob.OracleFsk(FS, RS, M, P=P, est_min=EST_MIN, est_max=EST_MAX)


def _cpu_worker(cpu, seconds, barrier, q):
    try:
        os.sched_setaffinity(0, {cpu})
    except Exception:
        pass
    ob, rx = _oracle_rx()
    buf = _CPU_BUF
    rx.demod(buf[:120_000], ob.IN_CU8_FSKDEMOD, want_filt=False, want_stats=False)      # library loaded, pages touched
    barrier.wait()
    done, t0 = 0, time.perf_counter()
    while True:
        r = rx.demod(buf, ob.IN_CU8_FSKDEMOD, want_filt=False, want_stats=False)
        done += r["consumed"]
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    q.put((done, dt))


def cpu_baseline(seconds):
    """CPU restatement (oracle, kind "port") timed on this host, on a bounded sample of the same workload. Called BEFORE the
    process touches the GPU (a fork of a process holding tens of GB of device mappings is what round 2 measured by mistake):
    one worker per usable core (len(os.sched_getaffinity(0)), not os.cpu_count(): a cgroup / affinity-limited box must not be
    over-counted), each pinned to its core, each demodulating the bench's 1.2 M-sample stream over and over. The workers
    load the oracle and warm up, meet at a barrier, and only then does each time ITS OWN demodulation loop for `seconds`
    of compute: start-up, fork and scheduling are outside every clock. value = sum of samples / the slowest worker's
    loop time; single_core_value = one pinned worker alone (run first)."""
    import multiprocessing as mp
    usable = sorted(os.sched_getaffinity(0))
    # a container may see every core of the host (affinity, cpu_count) and still be allowed only a few cores' worth of CPU
    # time by its cgroup (cpu.max): workers beyond that quota only time-slice -- round 3 measured 256 visible cores
    # delivering 8.2 cores of work. The baseline uses as many workers as the quota allows and says so.
    quota = cgroup_cpu_quota()
    if quota is not None and quota < len(usable):
        usable = usable[:max(1, int(quota))]
    cores = len(usable)
    ctx = mp.get_context("fork")

    def run(cpus, secs):
        bar = ctx.Barrier(len(cpus))
        q = ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker, args=(c, secs, bar, q)) for c in cpus]
        for p in procs:
            p.start()
        res = [q.get() for _ in procs]
        for p in procs:
            p.join()
        return res

    r1 = run(usable[:1], min(seconds, 6.0))
    res = run(usable, seconds)
    total = sum(r[0] for r in res)
    tmax = max(r[1] for r in res)
    rates = sorted(r[0] / r[1] / 1e6 for r in res)
    return {"value": total / tmax / 1e6, "unit": "IQ Msamples/s", "cores": cores, "kind": "port",
            "single_core_value": r1[0][0] / r1[0][1] / 1e6,
            "per_core_min_median_max": [rates[0], rates[len(rates) // 2], rates[-1]],
            "cgroup_cpu_quota": quota,
            "sample": f"{cores} pinned workers (sched_getaffinity: {len(os.sched_getaffinity(0))}, os.cpu_count: {os.cpu_count()}, cgroup cpu quota: "
                      f"{quota}), one oracle stream each, the "
                      f"bench's {len(_CPU_BUF) / 1e6:.2f} M-sample buffer demodulated repeatedly for {seconds:.0f} s of compute per core after a common barrier "
                      f"({total / 1e6:.0f} M samples in all, slowest loop {tmax:.2f} s); clocks inside the workers, around the "
                      f"demodulation loop only; single_core_value: one pinned worker alone"}


# ---- counter-derived annotations of the roofline object -------------------------------------------------------------------------------

def counter_annotations(h, lib, total_samples, kern_ms, M, TS, P, NSYM):
    """HBM bytes per launch and the VALU side of the headline kernel from the COMMITTED counter passes (profiles/hbm_traffic.json,
    collected in separate rocprofv3 --pmc runs), quoted only while the library that runs carries the kernel code object they were taken
    on. Returns (traffic_bytes_per_launch | None, traffic_source, valu | None, kernel_hash)."""
    import ctypes as C
    traffic, traffic_src, valu, khash = None, None, None, None
    try:
        lib.pirip_hip_kernel_source_hash.restype = C.c_char_p
        khash = lib.pirip_hip_kernel_source_hash().decode()
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if h.kernel() == "wave" and h.kernel_name() == tj.get("kernel_name", h.kernel_name()):
            fresh = tj.get("kernel_source_hash") == khash
            traffic_src = tj["source"] + ("" if fresh else f" -- STALE: taken on kernel object {tj.get('kernel_source_hash')}, this library is {khash}; "
                                                               "rerun tools/profile.sh and tools/update_hbm_traffic.py")
            if fresh:
                traffic = (tj["hbm_read_bytes_per_sample"] + tj["hbm_write_bytes_per_sample"]) * total_samples
                # the kernel is VALU-bound, not HBM-bound: report the instruction-issue side too, and how far the
                # executed instruction count is from what the arithmetic needs (tools/valu_floor.py walks the oracle's loop bounds)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import valu_floor
                fl = valu_floor.floor(M, TS, P, NSYM, 256, "u8")
                winst = tj["valu_instr_per_frame"] * (total_samples / (TS * NSYM)) / (kern_ms * 1e-3) / 1e9
                valu = {"achieved": winst, "peak": 614.4, "unit": "G wave64 VALU instr/s", "frac": winst / 614.4,
                        "instr_per_frame": tj["valu_instr_per_frame"], "source": tj["source"],
                        "floor_instr_per_frame": fl["floor_instr_per_frame"],
                        "executed_over_floor": tj["valu_instr_per_frame"] / fl["floor_instr_per_frame"],
                        # a figure that does not depend on the counters' 4-cycle unit: the arithmetic's mandatory lane operations
                        # (an fma counting once) per second against the FP32 vector peak, 157.3 TFLOP/s = 78.6 T lane-FMA/s
                        "useful_lane_ops_frac_of_fp32_peak": fl["floor_ops_per_sample"] * (total_samples / (kern_ms * 1e-3)) / 78.6e12,
                        "floor_lane_ops_per_sample": fl["floor_ops_per_sample"],
                        "floor_note": "tools/valu_floor.py: wave instructions the frame's arithmetic needs at perfect lane use and perfect "
                                      "f32 packing, estimator operations kept exactly as the oracle orders them (bit-exact Sf), "
                                      "correlator restructured as far as its tolerance allows; per phase: profiles/r04_phase_valu.txt",
                        "note": "peak = one wave64 instruction per SIMD per 4 cycles (the counters' unit); plain f32 ops issue "
                                "faster than that with >= 3 waves per SIMD, packed/DPP ops at ~2.8 cycles (profiles/r02_valu_issue.txt)"}
    except Exception as e:
        traffic_src = f"unavailable: {e!r}"
    return traffic, traffic_src, valu, khash


# ---- the checker: oracle replay of the device's bits ----------------------------------------------------------------------------------

def _check_worker(k):
    """Oracle replay of checked stream k: the device state has advanced `passes` passes over the same buffer. Returns the
    oracle's bits of the last pass and, per bit, whether the ORACLE's own decision was a near-tie (|mag0 - mag1| below
    NEAR_TIE of the peak: the two float32 evaluation orders may then legitimately decide differently)."""
    ob, rx = _oracle_rx()
    NSYM = _CFG[4]
    bufs, passes = _CHK
    ro = None
    for i in range(passes):
        ro = rx.demod(bufs[k], ob.IN_CU8_FSKDEMOD, want_filt=(i == passes - 1), want_stats=False)
    f = ro["rx_filt"]
    tie = np.abs(f[:, :NSYM] - f[:, NSYM:]) < NEAR_TIE * float(np.abs(f).max())
    return ro["bits"], tie


class Replayer:
    """Oracle replays of a fixed set of checked streams (host copies of their device buffers), one forked worker per core."""

    def __init__(self, bufs):
        self.bufs = bufs
        self.ncore = len(os.sched_getaffinity(0))
        q = cgroup_cpu_quota()
        if q is not None:
            self.ncore = max(1, min(self.ncore, int(q)))

    def replay(self, sel, hb, passes):
        """device bits hb[k] of checked streams sel after `passes` passes over the resident buffer, against the oracle and the sent frames:
        (bits differing outside near-ties, inside near-ties, errors vs the sent test frames, test bits, errors leaving out the first frame)"""
        import multiprocessing as mp
        from oracle import binding as ob
        global _CHK
        _CHK = (self.bufs[sel], passes)
        with mp.get_context("fork").Pool(min(self.ncore, max(len(sel), 1))) as pool:
            reps = pool.map(_check_worker, range(len(sel)))
        nbad = ntie = tx_err = tx_cnt = tx_err1 = 0
        for k, (obits, tie) in enumerate(reps):
            n = obits.shape[0]
            diff = hb[k, :n] != obits
            nbad += int((diff & ~tie).sum()); ntie += int((diff & tie).sum())
            res = ob.put_test_bits(hb[k, :n])                   # the DEVICE's bits against the transmitted test frames
            tx_err += res["errors"]; tx_cnt += res["bits"]
            tx_err1 += ob.put_test_bits(hb[k, 1:n])["errors"]   # ... leaving out the pass's first frame
        return nbad, ntie, tx_err, tx_cnt, tx_err1


def bit_checks(out, ctx):
    """The correctness side of the JSON line (rank 0, after the clock has stopped). ctx: the names bench.py's main() holds --
    torch, pirip_amd, h, dev, payloads, nstep, cons, maxf, nsamp, B, args, stream, dist, gather_out, frames_first, world."""
    from pirip_amd.shard import split_payload, unpack_bits
    torch, h, dev, payloads, nstep, cons = ctx["torch"], ctx["h"], ctx["dev"], ctx["payloads"], ctx["nstep"], ctx["cons"]
    maxf, nsamp, B, args, stream = ctx["maxf"], ctx["nsamp"], ctx["B"], ctx["args"], ctx["stream"]
    nchk = min(B, max(args.check_streams, 0))
    # streams strided across the WHOLE grid (first, last and evenly between): an addressing slip at high
    # workgroup indices must not hide behind a check of the first few streams
    idx = np.unique(np.linspace(0, B - 1, nchk).round().astype(np.int64)) if nchk else np.zeros(0, dtype=np.int64)
    tidx = torch.from_numpy(idx).cuda()
    last = payloads[(nstep - 1) % 2]
    if ctx["dist"]:
        # what rank 0 gathered: its own slot must be its own message, every rank must have delivered frames
        parts = [split_payload(g, B, maxf, h.Nbits) for g in ctx["gather_out"]]
        out["gather_check"] = {"rank0_echo": bool(torch.equal(ctx["gather_out"][0], last[0])),
                               "frames_per_rank": [int(p[1].sum()) for p in parts]}
    rp = Replayer(dev[tidx].cpu().numpy())
    # (1) the LAST TIMED step, whose demodulator state has been carried through warmup+steps passes over the same
    #     resident 1.2 M samples: bit for bit against an oracle that replays the same passes (a subset of the checked
    #     streams: the replay costs `passes` x the stream on a CPU core). Every pass restarts the recording under a
    #     demodulator that is mid-stream, so the first frames of a pass straddle a timing discontinuity and a few
    #     of their bits differ from what was SENT -- in the oracle exactly as on the device; that count is reported
    #     separately and is not the north star's "bit errors" figure.
    sel = np.unique(np.linspace(0, len(idx) - 1, min(len(idx), 32)).round().astype(np.int64)) if len(idx) else np.zeros(0, dtype=np.int64)
    hb_last = unpack_bits(last[1][tidx[torch.from_numpy(sel).cuda()]], h.Nbits).cpu().numpy() if len(sel) else np.zeros((0, 0, 0), dtype=np.uint8)
    nbad_t, ntie_t, tx_err_t, _, _ = rp.replay(sel, hb_last, args.warmup + args.steps)
    out["timed_step_check"] = {"streams": int(len(sel)), "passes_replayed": args.warmup + args.steps,
                               "bit_errors_vs_cpu_ref": nbad_t + ntie_t, "of_which_near_tie": ntie_t,
                               "bit_errors_vs_tx_incl_wraparound_frames": tx_err_t}
    # (2) one more, untimed, pass from the state fsk_create() leaves (pirip_hip_reset), i.e. the recording demodulated
    #     once from its start, as `fsk_demod` would: all checked streams against the oracle's single pass AND against
    #     the transmitted test frames (fsk_put_test_bits' count)
    h.reset(stream.cuda_stream)
    chk = payloads[nstep % 2]
    h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, chk[1].data_ptr(), maxf * chk[1].shape[2], 0, 0, 0, 0,
                  chk[2].data_ptr(), cons.data_ptr(), maxf, stream.cuda_stream)
    torch.cuda.synchronize()
    hb = unpack_bits(chk[1][tidx], h.Nbits).cpu().numpy()
    nbad, ntie, tx_err, tx_cnt, tx_err1 = rp.replay(np.arange(len(idx)), hb, 1)
    # A recording that starts mid-symbol hands the first decision of the first frame a fraction of a symbol: that bit can
    # differ from the SENT bit, in the oracle exactly as on the device (the only errors against the sent bits seen on this
    # noise-free workload). Against the oracle nothing differs: where the fraction is a single sample (a rounding tie at
    # -p 24) the first frame runs in the oracle's operation order (fsk_demod_exact0_kernel).
    out["bit_errors_vs_tx"] = tx_err
    out["bit_errors_vs_tx_after_first_frame"] = tx_err1
    out["ber_vs_tx"] = tx_err / max(tx_cnt, 1)
    # EVERY bit that differs from the CPU restatement's, whatever the reason; the near-tie class is a breakdown of it
    out["bit_errors_vs_cpu_ref"] = nbad + nbad_t + ntie + ntie_t
    out["near_tie_differences_vs_cpu_ref"] = {"count": ntie + ntie_t, "included_in_bit_errors_vs_cpu_ref": True,
                                              "rule": f"oracle's own |mag0 - mag1| < {NEAR_TIE} of the stream's peak"}
    out["bit_check"] = (f"{len(idx)} streams strided over all {B} (indices {int(idx[0]) if len(idx) else 0}..{int(idx[-1]) if len(idx) else 0}) "
                        f"x {ctx['frames_first']} frames: one untimed pass from the reset state vs the oracle and vs the tx test frames "
                        f"({tx_cnt} test bits); plus {len(sel)} of them on the last timed step vs an oracle replay of all "
                        f"{args.warmup + args.steps} passes (timed_step_check)")
    return {"rp": rp, "idx": idx, "tidx": tidx, "hb_reset_pass": hb}


# ---- side figures (N = 1, after the clock has stopped; none of them is `value`) -------------------------------------------------------

def _side_handle_rate(ctx, chk, make, what, reps=3, streams=None, samples=None):
    """The resident batch (or its first `streams` x `samples`) through another handle: its own output tensors (nothing of the measured
    run is overwritten), rate by HIP events, bits of the checked streams against the oracle replay of one pass from the reset state."""
    from pirip_amd.shard import alloc_payload, unpack_bits
    torch, dev, stream, pirip_amd = ctx["torch"], ctx["dev"], ctx["stream"], ctx["pirip_amd"]
    B = streams or ctx["B"]
    nsamp = samples or ctx["nsamp"]
    h2 = make(B)
    h2.set_bit_packing(True)
    maxf = h2.max_frames_for(nsamp)
    _, packed, nfr = alloc_payload(B, maxf, h2.Nbits, "cuda")
    cons2 = torch.zeros(B, dtype=torch.int64, device="cuda")
    run = lambda: h2.demod_batch(dev.data_ptr(), ctx["nsamp"] * 2, nsamp, packed.data_ptr(), maxf * packed.shape[2], 0, 0, 0, 0,
                                 nfr.data_ptr(), cons2.data_ptr(), maxf, stream.cuda_stream)
    run(); torch.cuda.synchronize()
    res = {"what": what, "kernel": h2.kernel_name(), "streams": B, "samples_per_stream": nsamp}
    if chk is not None and B == ctx["B"] and nsamp == ctx["nsamp"]:
        hbb = unpack_bits(packed[chk["tidx"]], h2.Nbits).cpu().numpy()
        nbad2, ntie2, tx_err2, tx_cnt2, _ = chk["rp"].replay(np.arange(len(chk["idx"])), hbb, 1)
        res.update({"bit_errors_vs_cpu_ref": nbad2 + ntie2, "bit_errors_vs_tx": tx_err2, "test_bits": tx_cnt2,
                    "bits_identical_to_the_measured_handle_on_the_checked_streams": bool(np.array_equal(hbb, chk["hb_reset_pass"]))})
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res.update({"kernel_ms": ms, "Msamples_per_s": float(cons2.sum()) / ms / 1e3})
    del h2
    return res


def side_figures(out, ctx, chk, khash, algo_bytes, hbm_peak):
    """(a) the OPT-IN band-only estimator (pirip_hip_set_estimator_band_only: Sf maintained only for the FFT bins the peak search of
    `fsk_demod --fsk_lower 500 --fsk_upper 25000` can read; every output identical); (b) PIRIP_KERNEL=exact, the mode that is word for
    word the CPU path at any SNR: its rate on a slice of the batch; (c) the other BASELINE configurations."""
    pirip_amd, local_rank = ctx["pirip_amd"], ctx["local_rank"]
    FS, RS, M, P, NSYM, EST_MIN, EST_MAX = _CFG
    mk = lambda n: pirip_amd.HipDemod(FS, RS, M, P=P, Nsym=NSYM, est_min=EST_MIN, est_max=EST_MAX, in_format=pirip_amd.IN_CU8_FSKDEMOD,
                                      nstreams=n, device=local_rank)
    try:
        def mk_band(n):
            h2 = mk(n)
            h2.set_estimator_band_only(True)
            return h2
        r = _side_handle_rate(ctx, chk, mk_band,
                              "pirip_hip_set_estimator_band_only(h, 1): Sf computed and smoothed for FFT bins 0..31 only (the peak search's "
                              "range at --fsk_lower 500 --fsk_upper 25000); default is the full estimator, which `value` is measured on")
        r["frac_of_hbm_roofline"] = r["Msamples_per_s"] * 1e6 * algo_bytes / 1e9 / hbm_peak
        if "bits_identical_to_the_measured_handle_on_the_checked_streams" in r:
            r["bits_identical_to_the_full_estimator_on_the_checked_streams"] = r.pop("bits_identical_to_the_measured_handle_on_the_checked_streams")
        try:   # its instruction count from the committed counter passes of this kernel build, against the floor of the pruned arithmetic
            tjb = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if tjb.get("kernel_source_hash") == khash and "band_only" in tjb:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import valu_floor
                TS = FS // RS
                flb = valu_floor.floor(M, TS, P, NSYM, 256, "u8", band_bins=32)["floor_instr_per_frame"]
                ipf = tjb["band_only"]["valu_instr_per_frame"]
                r.update({"valu_instr_per_frame": ipf, "floor_instr_per_frame": flb, "executed_over_floor": ipf / flb,
                          "hbm_bytes_per_sample": tjb["band_only"]["hbm_read_bytes_per_sample"] + tjb["band_only"]["hbm_write_bytes_per_sample"],
                          "counters": tjb["band_only"]["source"]})
        except Exception:
            pass
        out["opt_in_band_only_estimator"] = r
    except Exception as e:
        out["opt_in_band_only_estimator"] = f"unavailable: {e!r}"
    try:
        def mk_exact(n):
            os.environ["PIRIP_KERNEL"] = "exact"
            try:
                return mk(n)
            finally:
                del os.environ["PIRIP_KERNEL"]
        r = _side_handle_rate(ctx, None, mk_exact,
                              "PIRIP_KERNEL=exact: every frame in the CPU restatement's own operation order (one thread walks the oscillator and "
                              "timing sums): every output word equal to the CPU path at any SNR -- the mode for a user who needs that under noise; "
                              "never the default, never `value`", reps=1, streams=min(ctx["B"], 4096), samples=min(ctx["nsamp"], 120_000))
        r["of_the_default_kernels_rate"] = None
        out["exact_order_kernel"] = r
    except Exception as e:
        out["exact_order_kernel"] = f"unavailable: {e!r}"


def extra_configs(out, iters=3):
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        out["extra_configs"] = bench_configs.measure(iters=iters)
    except Exception as e:
        out["extra_configs"] = f"unavailable: {e!r}"
