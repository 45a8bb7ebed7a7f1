#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s demodulated (2-FSK Fs=240k Rs=10k), BASELINE.json's metric.

A "step" is one pass of the hot path (u8 IQ -> bits, `fsk_demod -d -p 24 2 240000 10000`,
/root/reference/README.md:105) over one batch of B independent synthetic IQ streams that are
already resident in HBM (BASELINE config 2). One process per GPU; streams shard
one block per rank with no data-path collective; each step ends with ONE gather of the decoded
bits to rank 0 over RCCL (torch.distributed backend "nccl"), inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B] [--samples S]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work at every N: started WITHOUT a torchrun environment (no WORLD_SIZE) and asked for N > 1 (or for
--exercise-gather), the script counts the visible GPUs, refuses if there are fewer than N ("N GPUs requested, V visible"),
and otherwise replaces itself (exec) with the torch.distributed.run command line above on a free port of 127.0.0.1 --
the driver's N = 1 invocation and an N = 8 invocation are then the same command with a different number.

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md "Measurement"). This file is the timed path, top to bottom: synthetic
batch -> handle -> step() -> clock -> JSON; the checker and the side figures live in tools/bench_checks.py, the launcher logic in
tools/bench_launch.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# how this process starts (rank / single run / launcher of N ranks) and how the CPU leg reaches rank 0: tools/bench_launch.py;
# everything reported beside the timed region (CPU leg, oracle replay, counter annotations, side figures): tools/bench_checks.py
import bench_checks                                                                                                  # noqa: E402
from bench_launch import (CPU_ENV, clear_stale_cpu_leg, cpu_flag_path, cpu_leg_plan, free_port, launch_plan,         # noqa: E402,F401
                          launcher_argv, publish_cpu_leg, visible_gpus, wait_for_cpu_leg)

FS, RS, M, P, NSYM = 240000, 10000, 2, 24, 50
TS = FS // RS
EST_MIN, EST_MAX = 500, 25000
F1, SHIFT = 10000, 10000
# u8 I + u8 Q read; decoded bits written packed 8 per byte, 7 bytes per 1200-sample frame ("if bits are packed
# ... state which") -- the output mode of EVERY N, so the points of a scaling curve are the same work per GPU
ALGO_BYTES_PER_SAMPLE = 2.0 + 7.0 / 1200.0
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


def main():
    cfg = (FS, RS, M, P, NSYM, EST_MIN, EST_MAX)
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="compute time per core of the CPU leg")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="time the CPU leg on this host, print its JSON object and exit (no GPU touched): what the "
                         "self-launcher runs before it becomes torch.distributed.run")
    ap.add_argument("--cpu-leg-only", action="store_true",
                    help="run this rank's part of the CPU-leg hand-over (measure / wait / quote) and exit: CPU test hook")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        base, _ = synth_base_streams(args.samples)
        bench_checks.configure(cfg, np.ascontiguousarray(base[2][:args.samples]))
        print(json.dumps(bench_checks.cpu_baseline(args.cpu_seconds)), flush=True)
        return

    what, why = launch_plan(args.gpus, args.exercise_gather, os.environ, visible_gpus)
    if what == "refuse":
        raise SystemExit(f"bench.py: {why}")
    if what == "spawn":
        argv = launcher_argv(args.gpus, free_port(), os.path.abspath(__file__), sys.argv[1:])
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", "1")
        if not args.no_cpu_baseline and args.ebno_db is None and CPU_ENV not in env:
            # the CPU leg of an N-rank job: timed NOW, while this is the only process of the job, in a child of its own (this
            # process has already asked the HIP runtime for the device count), and handed to rank 0 through the environment
            import subprocess
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-seconds",
                                str(args.cpu_seconds), "--samples", str(args.samples)], capture_output=True, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            env[CPU_ENV] = lines[-1] if r.returncode == 0 and lines else json.dumps({"error": (r.stderr or "no output")[-400:]})
        print("bench.py: starting " + " ".join(argv[1:]), file=sys.stderr, flush=True)
        os.execve(argv[0], argv, env)                          # this process BECOMES the launcher: one JSON line, one exit code

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    B, nsamp = args.streams, args.samples
    cpu_how, cpu_arg = cpu_leg_plan(rank, world, os.environ, args.no_cpu_baseline or args.ebno_db is not None)
    waited = None
    if cpu_how == "wait":                             # rank 0 is timing the oracle on this host's cores: stay off them
        waited = wait_for_cpu_leg(cpu_arg, args.cpu_seconds * 2 + 180.0)
    base, txbits = synth_base_streams(nsamp)          # CPU Tx side; nothing here touches the GPU
    bench_checks.configure(cfg, np.ascontiguousarray(base[2][:nsamp]))
    cpu_res = None
    if cpu_how == "env":
        try:
            cpu_res = json.loads(cpu_arg)
            cpu_res["timed"] = "by bench.py's launcher process before it started the ranks (same run, same host, no rank alive yet)"
        except Exception as e:
            cpu_res = {"error": f"{CPU_ENV}: {e!r}"}
    elif cpu_how == "measure":
        if cpu_arg:
            clear_stale_cpu_leg(cpu_arg)               # (a crashed job with this parent pid and port must not end the others' wait early)
        try:
            cpu_res = bench_checks.cpu_baseline(args.cpu_seconds)   # before any device allocation: see cpu_baseline()
            if world > 1:
                cpu_res["timed"] = "by rank 0 before any rank touched its GPU; the other ranks slept until it finished"
        except Exception as e:
            cpu_res = {"error": repr(e)}
        if cpu_arg:
            publish_cpu_leg(cpu_arg, cpu_res)
    if args.cpu_leg_only:
        print(json.dumps({"rank": rank, "how": cpu_how, "waited_s": waited, "cpu_baseline": cpu_res}), flush=True)
        if cpu_how == "measure" and cpu_arg:
            time.sleep(0.5)
            os.unlink(cpu_arg)
        return

    import torch
    import pirip_amd
    if not torch.cuda.is_available() or pirip_amd.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible")
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
        if cpu_how == "measure" and cpu_arg:           # every rank is past its wait once the rendezvous has completed
            try:
                os.unlink(cpu_arg)
            except OSError:
                pass

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
        bench_checks.configure(cfg, dev[2].cpu().numpy())

    h = pirip_amd.HipDemod(FS, RS, M, P=P, Nsym=NSYM, est_min=EST_MIN, est_max=EST_MAX,
                           in_format=pirip_amd.IN_CU8_FSKDEMOD, nstreams=B, device=local_rank)
    maxf = h.max_frames_for(nsamp)
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    from pirip_amd.shard import alloc_payload, gather_payload, split_payload, unpack_bits
    gather_out = None
    works = []
    # EVERY N: the kernel emits packed bits (8 per byte) and frame counts straight into the gather message (the same output
    # mode at N = 1 and N = 8); two messages alternate so that at N > 1 a gather in flight never aliases the next launch's output
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
        while len(works) > 1:                   # the message about to be rewritten was sent two steps ago
            works.pop(0)[0].wait()
        payload, packed, nfr_p = payloads[nstep % 2]
        h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, packed.data_ptr(), maxf * packed.shape[2], 0, 0, 0, 0,
                      nfr_p.data_ptr(), cons.data_ptr(), maxf, stream.cuda_stream)
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
    frames_first = int(payloads[(nstep - 1) % 2][2][0])
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
    # every rank's own kernel time and sample count, so that rank 0 can print one roofline per GPU
    per_rank = torch.tensor([[kern_ms, float(cons.sum())]], dtype=torch.float64, device="cuda")
    if dist:
        allr = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
        per_rank = torch.cat(allr)
    per_rank = per_rank.cpu().numpy()
    rccl_world = dist.get_world_size() if dist else None
    rccl_backend = dist.get_backend() if dist else None

    if rank == 0:
        total = float(cons.sum())
        value = samples_per_step * args.steps / dt / 1e6
        ach = (total * ALGO_BYTES_PER_SAMPLE) / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src, valu, khash = bench_checks.counter_annotations(h, pirip_amd.lib(), total, kern_ms, M, TS, P, NSYM)
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
                                       "1 GPU: all streams on it; bits written packed into the gather message as at N > 1, no exchange at N = 1"),
                       "output": "packed bits, 7 bytes per 50-bit frame, + int32 frame counts, written in place into the gather message",
                       "kernel": h.kernel_name()},   # pirip_hip_get_kernel_name(): what the handle actually runs
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": (traffic_src or "") + " (committed PMC profile scaled by this run's samples, not a counter read of the timed run)",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE},
            "valu": valu,
            # one entry per rank = per GPU: its own HIP-event kernel time and the roofline fraction that follows from it
            "per_gpu": [{"rank": r, "kernel_ms": float(per_rank[r, 0]),
                         "achieved_GBps": float(per_rank[r, 1]) * ALGO_BYTES_PER_SAMPLE / (float(per_rank[r, 0]) * 1e-3) / 1e9,
                         "frac": float(per_rank[r, 1]) * ALGO_BYTES_PER_SAMPLE / (float(per_rank[r, 0]) * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                        for r in range(per_rank.shape[0])],
            "rccl": ({"world_size": rccl_world, "backend": rccl_backend, "exchange": "one gather of the packed-bit message to rank 0 per step"}
                     if dist else None),
        }
        # everything below runs after the clock has stopped: the checker (oracle replay), the CPU leg's figure, the side figures
        ctx = {"torch": torch, "pirip_amd": pirip_amd, "h": h, "dev": dev, "payloads": payloads, "nstep": nstep, "cons": cons, "maxf": maxf,
               "nsamp": nsamp, "B": B, "args": args, "stream": stream, "dist": dist, "gather_out": gather_out, "frames_first": frames_first,
               "world": world, "local_rank": local_rank}
        chk = None
        try:
            chk = bench_checks.bit_checks(out, ctx)
        except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
            out["bit_check"] = f"unavailable: {e!r}"
        if world == 1 and not args.no_extra:
            bench_checks.side_figures(out, ctx, chk, khash, ALGO_BYTES_PER_SAMPLE, HBM_PEAK_GBPS)
            if isinstance(out.get("exact_order_kernel"), dict):
                out["exact_order_kernel"]["of_the_default_kernels_rate"] = out["exact_order_kernel"]["Msamples_per_s"] / value
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        if world == 1 and not args.no_extra:
            # the other BASELINE configurations, as side keys with their own roofline fractions (not the metric)
            del dev, payloads, ctx, chk
            torch.cuda.empty_cache()
            bench_checks.extra_configs(out, iters=3)
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
