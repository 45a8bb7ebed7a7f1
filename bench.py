#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated (2-FSK Fs=240k Rs=10k), BASELINE.json's metric.

A "step" is one pass of the hot path (u8 IQ -> bits, `fsk_demod -d -p 24 2 240000 10000`,
/root/reference/README.md:105) over one batch of B independent synthetic IQ streams that are
already resident in HBM (BASELINE config 2). One process per GPU; streams shard
one block per rank with no data-path collective; each step ends with ONE gather of the decoded
bits to rank 0 over RCCL (torch.distributed backend "nccl"), inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B] [--samples S]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, RS, M, P, NSYM = 240000, 10000, 2, 24, 50
TS = FS // RS
EST_MIN, EST_MAX = 500, 25000
F1, SHIFT = 10000, 10000
ALGO_BYTES_PER_SAMPLE = 2.0 + 1.0 / 24.0     # u8 I + u8 Q read, one byte per decoded bit written
HBM_PEAK_GBPS = 8000.0                        # MI355X spec (MI355X_MICROARCH.md)
N_PLANS = 5                                   # distinct tone plans (k * 937.5 Hz shifts)


def synth_base_streams(nsamp):
    """Product-side synthetic Tx (no oracle): test bits -> fsk_mod_c (libpirip_hip's codec2-shim
    modulator, CPU Tx side) -> u8 quantiser u8 = clamp(rint(127 + 32*x)). Returns
    [N_PLANS, nsamp + TS, 2] uint8: plan k has its tones shifted by (k-2) bins of 937.5 Hz."""
    import ctypes as C
    import pirip_amd
    L = pirip_amd.lib()
    L.fsk_create_hbr.restype = C.c_void_p
    L.fsk_create_hbr.argtypes = [C.c_int] * 7
    L.fsk_mod_c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fsk_destroy.argtypes = [C.c_void_p]
    bin_path = os.path.join(ROOT, "pirip_amd", "bin", "fsk_get_test_bits")
    import subprocess
    nsym = (nsamp + TS) // TS + NSYM
    nsym -= nsym % NSYM
    bits = np.frombuffer(subprocess.run([bin_path, "-", str(nsym)], capture_output=True, check=True).stdout,
                         dtype=np.uint8)[:nsym].copy()
    out = np.zeros((N_PLANS, nsamp + TS, 2), dtype=np.uint8)
    for k in range(N_PLANS):
        f1 = F1 + int(round((k - 2) * 937.5))
        fsk = L.fsk_create_hbr(FS, RS, M, P, NSYM, f1, SHIFT)
        x = np.zeros((nsym * TS, 2), dtype=np.float32)
        for i in range(0, nsym, NSYM):          # 50 symbols per fsk_mod_c call, like the fsk_mod tool
            seg = x[i * TS:(i + NSYM) * TS]
            L.fsk_mod_c(fsk, seg.ctypes.data, bits[i:i + NSYM].ctypes.data, NSYM)
        L.fsk_destroy(fsk)
        q = np.clip(np.rint(127.0 + 32.0 * x[:nsamp + TS].astype(np.float64)), 0, 255).astype(np.uint8)
        out[k] = q
    return out, bits


def cpu_baseline(sample_samples):
    """CPU restatement (oracle, kind "port") timed on this host, on a bounded sample of the same workload: (1) ONE process
    pinned to one core -- the single-core rate; (2) one stream per usable core (len(os.sched_getaffinity(0)), not
    os.cpu_count(): a cgroup / affinity-limited box must not be over-counted), all at once -- the whole-host rate."""
    import multiprocessing as mp
    usable = sorted(os.sched_getaffinity(0))
    cores = len(usable)
    ctx = mp.get_context("fork")
    with ctx.Pool(1, initializer=_pin, initargs=(usable[:1],)) as pool:
        t0 = time.time()
        done1 = pool.map(_cpu_worker, [max(sample_samples // 2, 1_200_000)])[0]
        dt1 = time.time() - t0
    with ctx.Pool(cores) as pool:
        t0 = time.time()
        res = pool.map(_cpu_worker, [sample_samples] * cores)
        dt = time.time() - t0
    total = sum(r for r in res)
    return {"value": total / dt / 1e6, "unit": "IQ Msamples/s", "cores": cores, "kind": "port",
            "single_core_value": done1 / dt1 / 1e6,
            "sample": f"{cores} streams x {sample_samples} samples of the bench workload, one oracle stream per usable core "
                      f"(sched_getaffinity: {cores}, os.cpu_count: {os.cpu_count()}), wall {dt:.1f} s; single_core_value: one pinned "
                      f"process, {done1} samples in {dt1:.1f} s"}


def _pin(cpus):
    try:
        os.sched_setaffinity(0, set(cpus))
    except Exception:
        pass


_CPU_BUF = None
_CHK = None


def _check_worker(k):
    """Oracle replay of checked stream k: the device state has advanced warmup+steps passes over the same buffer."""
    from oracle import binding as ob
    bufs, passes = _CHK
    rx = ob.OracleFsk(FS, RS, M, P=P, est_min=EST_MIN, est_max=EST_MAX)
    ro = None
    for _ in range(passes):
        ro = rx.demod(bufs[k], ob.IN_CU8_FSKDEMOD, want_filt=False, want_stats=False)
    return ro["bits"], ob.put_test_bits(ro["bits"])


def _cpu_worker(nsamp):
    from oracle import binding as ob
    rx = ob.OracleFsk(FS, RS, M, P=P, est_min=EST_MIN, est_max=EST_MAX)
    buf = _CPU_BUF
    done = 0
    while done < nsamp:
        r = rx.demod(buf, ob.IN_CU8_FSKDEMOD, want_filt=False, want_stats=False)
        done += r["consumed"]
    return done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=16384,
                    help="streams per GPU (weak scaling); 16384 x 1.2 M samples = 39 GB of u8 IQ resident in HBM. "
                         "Measured on one MI355X (round 2, 3 waves per SIMD = 3072 resident streams): 6144 -> 256 G samples/s, "
                         "15360 -> 260, 16384 -> 260")
    ap.add_argument("--samples", type=int, default=1_200_000, help="IQ samples per stream per step")
    ap.add_argument("--ebno-db", type=float, default=None,
                    help="regenerate the batch with the device-side Tx (pirip_hip_synth_cu8) and AWGN at this Eb/N0; "
                         "default: noise-free fsk_mod IQ (the BASELINE workload)")
    ap.add_argument("--exercise-gather", action="store_true",
                    help="run the N>1 code path (in-place packed message + RCCL gather) even at world size 1")
    ap.add_argument("--check-streams", type=int, default=256,
                    help="streams of rank 0, strided across the whole batch, compared bit for bit with an oracle replay")
    ap.add_argument("--no-extra", action="store_true", help="skip the config 3 / config 4 side measurements (N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=24_000_000, help="samples per core for the CPU leg")
    args = ap.parse_args()

    import torch
    import pirip_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available() or pirip_amd.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.exercise_gather:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    B, nsamp = args.streams, args.samples
    base, txbits = synth_base_streams(nsamp)
    global _CPU_BUF
    _CPU_BUF = np.ascontiguousarray(base[2][:nsamp])

    # device-resident batch: stream s = plan (s % N_PLANS), timing offset (s // N_PLANS) % TS samples
    dbase = torch.from_numpy(base).cuda()
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    period = N_PLANS * TS                       # distinct (plan, offset) combinations
    for c in range(min(period, B)):
        gs = rank * B + c                       # global index of the first stream with this combination
        src = dbase[gs % N_PLANS, (gs // N_PLANS) % TS:(gs // N_PLANS) % TS + nsamp]
        dev[c::period] = src.unsqueeze(0)       # streams c, c+period, ... share (plan, offset) when B % period == 0
    if (rank * B) % period or B % period:
        for s in range(B):                      # general case: per-stream copies
            gs = rank * B + s
            dev[s].copy_(dbase[gs % N_PLANS, (gs // N_PLANS) % TS:(gs // N_PLANS) % TS + nsamp])
    del dbase
    if args.ebno_db is not None:
        # same (plan, offset) assignment, but modulated and noised on the device, every stream its own noise
        from pirip_amd.binding import synth_cu8
        gsi = rank * B + np.arange(B)
        f1s = F1 + np.rint(((gsi % N_PLANS) - 2) * 937.5).astype(np.int32)
        skips = ((gsi // N_PLANS) % TS).astype(np.int32)
        sigma = float(np.sqrt((4.0 * TS / 10 ** (args.ebno_db / 10.0)) / 2.0))
        amp = 16.0 if args.ebno_db >= 10 else 8.0
        dtx = torch.from_numpy(txbits).cuda()
        synth_cu8(FS, RS, M, f1s, SHIFT, dtx.data_ptr(), 0, int(txbits.size), dev.data_ptr(), nsamp * 2, nsamp,
                  amp=amp, sigma=sigma, seed=0x5eed + rank, skip=skips, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        _CPU_BUF = dev[2].cpu().numpy()

    h = pirip_amd.HipDemod(FS, RS, M, P=P, Nsym=NSYM, est_min=EST_MIN, est_max=EST_MAX,
                           in_format=pirip_amd.IN_CU8_FSKDEMOD, nstreams=B, device=local_rank)
    maxf = h.max_frames_for(nsamp)
    bits = torch.zeros((B, maxf, h.Nbits), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    from pirip_amd.shard import alloc_payload, gather_payload, split_payload
    gather_out = None
    works = []
    payloads = None
    if dist:
        # N>1: the kernel emits packed bits (8 per byte) and frame counts straight into the gather message;
        # two messages alternate so a gather in flight never aliases the next launch's output
        h.set_bit_packing(True)
        payloads = [alloc_payload(B, maxf, h.Nbits, "cuda") for _ in range(2)]
    nstep = 0
    stream = torch.cuda.current_stream()

    kev = []

    def step(timed):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        nonlocal gather_out, nstep
        if dist:
            while len(works) > 1:               # the message about to be rewritten was sent two steps ago
                works.pop(0)[0].wait()
            payload, packed, nfr_p = payloads[nstep % 2]
            h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, packed.data_ptr(), maxf * packed.shape[2], 0, 0, 0, 0,
                          nfr_p.data_ptr(), cons.data_ptr(), maxf, stream.cuda_stream)
        else:
            h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * h.Nbits, 0, 0, 0, 0,
                          nfr.data_ptr(), cons.data_ptr(), maxf, stream.cuda_stream)
        if timed:
            e1.record(stream)
            kev.append((e0, e1))
        nstep += 1
        if dist:
            # the single RCCL exchange of the path: decoded bits packed 8 per byte + frame counts in ONE
            # gather to rank 0, asynchronous on RCCL's stream so it overlaps the next step's kernel
            if rank == 0 and gather_out is None:
                gather_out = [torch.empty_like(payload) for _ in range(world)]
            _, work = gather_payload(payload, dist, rank, world, 0, gather_out, async_op=True)
            works.append((work, payload))

    for _ in range(args.warmup):
        step(False)
    while works:
        works.pop(0)[0].wait()
    torch.cuda.synchronize()
    # correctness gate on the warm-up output (rank 0, a few streams): decoded bits must be the
    # transmitted test frames -- 0 errors -- before any number is reported
    frames_first = int((payloads[(nstep - 1) % 2][2] if dist else nfr)[0])
    consumed_total = int(cons.sum())
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    while works:
        works.pop(0)[0].wait()                  # every gather has landed before the clock stops
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # per-step consumed samples (identical every step up to +-1 frame per stream)
    cons_step = torch.tensor([float(cons.sum())], dtype=torch.float64, device="cuda")
    if dist:
        dist.all_reduce(cons_step, op=dist.ReduceOp.SUM)
    samples_per_step = float(cons_step.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))

    if rank == 0:
        value = samples_per_step * args.steps / dt / 1e6
        ach = (float(cons.sum()) * ALGO_BYTES_PER_SAMPLE) / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src, valu = None, None, None
        try:   # HBM bytes per launch from the committed PMC passes (collected in separate rocprofv3 --pmc runs)
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if not (os.environ.get("PIRIP_FORCE_GENERAL") or os.environ.get("PIRIP_KERNEL") == "general"):
                traffic = (tj["hbm_read_bytes_per_sample"] + tj["hbm_write_bytes_per_sample"]) * float(cons.sum())
                traffic_src = tj["source"]
                # the kernel is VALU-bound, not HBM-bound: report the instruction-issue side too
                winst = tj["valu_instr_per_frame"] * (float(cons.sum()) / (TS * NSYM)) / (kern_ms * 1e-3) / 1e9
                valu = {"achieved": winst, "peak": 614.4, "unit": "G wave64 VALU instr/s", "frac": winst / 614.4,
                        "instr_per_frame": tj["valu_instr_per_frame"], "source": tj["source"],
                        "note": "peak = one wave64 instruction per SIMD per 4 cycles (the counters' unit); plain f32 ops issue "
                                "faster than that with >= 3 waves per SIMD, packed/DPP ops at ~2.8 cycles (profiles/r02_valu_issue.txt)"}
        except Exception:
            pass
        out = {
            "metric": "IQ Msamples/s demodulated (2-FSK Fs=240k Rs=10k); BER vs CPU ref",
            "value": value, "unit": "IQ Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 2-FSK Fs=240k Rs=10k -p 24, batched synthetic u8 IQ, "
                                   "device-resident (fsk_demod -d equivalent)"
                                   + ("" if args.ebno_db is None else f", device-side Tx with AWGN at Eb/N0 {args.ebno_db} dB"),
                       "streams_per_gpu": B, "samples_per_stream": nsamp, "frames_per_stream": frames_first,
                       "parallelism": (f"streams sharded {world}x, one RCCL gather of packed bits per step" if dist else
                                       "1 GPU: all streams on it, no gather at N = 1 (one byte per bit written to HBM)"),
                       "kernel": "fsk_demod_general" if (os.environ.get("PIRIP_FORCE_GENERAL") or os.environ.get("PIRIP_KERNEL") == "general")
                                 else "fsk_demod_wave_kernel<2,24,24,50,256,u8 -d,4 streams/block,3 waves/SIMD>"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": (traffic_src or "") + " (committed PMC profile scaled by this run's samples, not a counter read of the timed run)",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE},
            "valu": valu,
        }
        # bit check + CPU baseline (rank 0, N=1 only for the baseline)
        try:
            from oracle import binding as ob  # noqa: F401  (checker only)
            import multiprocessing as mp
            nchk = min(B, max(args.check_streams, 0))
            # streams strided across the WHOLE grid (first, last and evenly between): an addressing slip at high
            # workgroup indices must not hide behind a check of the first few streams
            idx = np.unique(np.linspace(0, B - 1, nchk).round().astype(np.int64)) if nchk else np.zeros(0, dtype=np.int64)
            tidx = torch.from_numpy(idx).cuda()
            if dist:
                from pirip_amd.shard import unpack_bits
                last = payloads[(nstep - 1) % 2]
                hb = unpack_bits(last[1][tidx], h.Nbits).cpu().numpy()
                # what rank 0 gathered: its own slot must be its own message, every rank must have delivered frames
                parts = [split_payload(g, B, maxf, h.Nbits) for g in gather_out]
                out["gather_check"] = {"rank0_echo": bool(torch.equal(gather_out[0], last[0])),
                                       "frames_per_rank": [int(p[1].sum()) for p in parts]}
            else:
                hb = bits[tidx].cpu().numpy()
            bufs = dev[tidx].cpu().numpy()
            global _CHK
            _CHK = (bufs, args.warmup + args.steps)
            with mp.get_context("fork").Pool(min(len(os.sched_getaffinity(0)), max(len(idx), 1))) as pool:
                reps = pool.map(_check_worker, range(len(idx)))
            nbad = tx_err = tx_cnt = 0
            for k, (obits, res) in enumerate(reps):
                n = obits.shape[0]
                nbad += int((hb[k, :n] != obits).sum())
                tx_err += res["errors"]; tx_cnt += res["bits"]
            # vs the CPU reference: every decoded bit of the checked streams; vs the transmitted test frames:
            # fsk_put_test_bits' count (noise-free it must be 0 once the estimators have settled)
            out["bit_errors_vs_tx"] = tx_err
            out["ber_vs_tx"] = tx_err / max(tx_cnt, 1)
            out["bit_errors_vs_cpu_ref"] = nbad
            out["bit_check"] = (f"{len(idx)} streams strided over all {B} (indices {int(idx[0]) if len(idx) else 0}..{int(idx[-1]) if len(idx) else 0}) "
                                f"x {frames_first} frames of the last step vs oracle replay, and vs tx test frames")
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
            out["bit_check"] = f"unavailable: {e!r}"
        if world == 1 and not args.no_extra:
            # the other BASELINE configurations, as side keys with their own roofline fractions (not the metric)
            try:
                del dev, bits
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_configs
                out["extra_configs"] = bench_configs.measure(iters=3)
            except Exception as e:
                out["extra_configs"] = f"unavailable: {e!r}"
        line = json.dumps(out)
    if dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the last line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
