#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations on one GPU (profiling aid, not the driver's bench):
  config 3: u8 IQ at 1.8 MS/s -> /45 decimator -> cs16 -> 2-FSK Rs=1k demod at 40 kS/s (general kernel)
  config 4: 4-FSK Fs=240k Rs=10k demod half (fast kernel instance M=4 P=8; LDPC half is blocked)
Synthetic streams come from the product's CPU modulator (no oracle); a few streams are checked
against the oracle afterwards when it is available."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def modulate(L, Fs, Rs, M, f1, shift, nsym, seed, bits=None):
    L.fsk_create_hbr.restype = C.c_void_p
    L.fsk_create_hbr.argtypes = [C.c_int] * 7
    L.fsk_mod_c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fsk_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(seed)
    bps = 1 if M == 2 else 2
    if bits is not None:
        nsym = len(bits) // bps
    nsym -= nsym % 50
    if bits is None:
        bits = rng.integers(0, 2, nsym * bps).astype(np.uint8)
    bits = np.ascontiguousarray(bits[:nsym * bps], dtype=np.uint8)
    Ts = Fs // Rs
    fsk = L.fsk_create_hbr(Fs, Rs, M, 8 if Ts % 8 == 0 else Ts, 50, f1, shift)    # (the modulator does not use P; it must divide Ts)
    x = np.zeros((nsym * Ts, 2), dtype=np.float32)
    for i in range(0, nsym, 50):
        seg = x[i * Ts:(i + 50) * Ts]
        L.fsk_mod_c(fsk, seg.ctypes.data, bits[i * bps:(i + 50) * bps].ctypes.data, 50 * bps)
    L.fsk_destroy(fsk)
    return x, bits


def measure(iters=5):
    class A: pass
    args = A(); args.iters = iters
    import torch
    import pirip_amd
    L = pirip_amd.lib()
    st = torch.cuda.current_stream()
    res = {}

    # ---- config 4: 4-FSK, 2048 streams x 600k samples ----------------------------------------
    B, nsamp = 8192, 600_000
    x, _ = modulate(L, 240000, 10000, 4, 10000, 10000, nsamp // 24 + 50, 1)
    u8 = np.clip(np.rint(127.0 + 32.0 * x[:nsamp + 24].astype(np.float64)), 0, 255).astype(np.uint8)
    d = torch.from_numpy(u8).cuda()
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    for c in range(24):
        dev[c::24] = d[c:c + nsamp].unsqueeze(0)
    h = pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=B)
    maxf = h.max_frames_for(nsamp)
    bits = torch.zeros((B, maxf, 100), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    run = lambda: h.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * 100, 0, 0, 0, 0,
                                nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.iters):
        run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    c4 = float(cons.sum()) / ms / 1e3
    res["config4_4fsk_demod"] = {"workload": "BASELINE configs[3], demodulator half: 4-FSK Fs=240k Rs=10k P=8, u8 IQ, device-resident",
                                 "streams": B, "samples_per_stream": nsamp, "kernel_ms": ms, "Msamples_per_s": c4,
                                 "roofline": {"bound": "hbm", "achieved": c4 * 1e6 * (2.0 + 100.0 / 1200.0) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                              "frac": c4 * 1e6 * (2.0 + 100.0 / 1200.0) / 1e9 / 8000.0,
                                              "algorithmic_bytes_per_sample": 2.0 + 100.0 / 1200.0}}
    # the same batch with the OPT-IN band-only estimator (pirip_hip_set_estimator_band_only: Sf for FFT bins 0..63, what a 500..60000 Hz
    # peak search can read; outputs identical): side figure, the entry's own numbers above are the full estimator's
    try:
        hb = pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=B)
        hb.set_estimator_band_only(True)
        bits2 = torch.zeros((B, maxf, 100), dtype=torch.uint8, device="cuda")
        runb = lambda: hb.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits2.data_ptr(), maxf * 100, 0, 0, 0, 0,
                                      nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
        h.reset(); run(); hb.reset(); runb(); torch.cuda.synchronize()
        same = bool(torch.equal(bits, bits2))
        e0.record(st)
        for _ in range(args.iters):
            runb()
        e1.record(st); torch.cuda.synchronize()
        msb = e0.elapsed_time(e1) / args.iters
        res["config4_4fsk_demod"]["opt_in_band_only_estimator"] = {"kernel": hb.kernel_name(), "kernel_ms": msb, "Msamples_per_s": float(cons.sum()) / msb / 1e3,
                                                                   "bits_identical_to_the_full_estimator": same}
        del hb, bits2
    except Exception as e:
        res["config4_4fsk_demod"]["opt_in_band_only_estimator"] = f"unavailable: {e!r}"
    del dev, bits

    # ---- config 4, whole receive chain: 4-FSK demod with soft decisions -> FSK_LDPC receive (LLRs, UW sync, decode, CRC16) --
    # continuous test frames from the repo's framer (stand-in rate-1/2 (512,256) code), Eb/N0 ~ 7 dB, every stream the same
    # signal at its own sample offset; counts the frames delivered with a good CRC
    import subprocess
    B, nsamp = 8192, 600_000
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = subprocess.run([framer, "--code", pirip_amd.STANDIN_CODE, "--testframes", "93", "--seq", "--source", "0x1", "/dev/zero", "-"],
                        capture_output=True, check=True).stdout
    x, _ = modulate(L, 240000, 10000, 4, 10000, 10000, 0, 3, bits=np.frombuffer(fb, dtype=np.uint8))
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    h4 = pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=B)
    maxf = h4.max_frames_for(nsamp)
    filt = torch.zeros((B, maxf, 200), dtype=torch.float32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    ld = pirip_amd.HipLdpc(pirip_amd.STANDIN_CODE, 4, nstreams=B)
    stt = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda")
    pay = torch.zeros((B, maxf, 32), dtype=torch.uint8, device="cuda")
    inf = torch.zeros((B, maxf, pirip_amd.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")

    def dem():
        h4.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, 0, 0, filt.data_ptr(), maxf * 200, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)

    def dec():
        ld.rx_batch(filt.data_ptr(), maxf * 200, nfr.data_ptr(), maxf, stt.data_ptr(), pay.data_ptr(), inf.data_ptr(), st.cuda_stream)
    # algorithmic bytes per IQ sample: 2 read + the bits out (100 per 1200 samples); the soft decisions / LLRs the
    # two stages hand each other are INTERMEDIATE traffic, reported beside it, not part of the roofline figure
    ab4 = 2.0 + 100.0 / 1200.0
    # fused hand-over: bit LLRs written once (400 B per 1200 samples) and read once by the decoder, hard-decision words, records
    inter4 = 2.0 * 400.0 / 1200.0 + 2.0 * 16.0 / 1200.0 + (1 + 32 + 40) / 1200.0
    for ebno_db, key in ((7.0, "config4_4fsk_demod_plus_ldpc"), (3.5, "config4_4fsk_demod_plus_ldpc_low_snr")):
        rng = np.random.default_rng(5)
        sigma = np.sqrt((4.0 * 24 / 2.0) / (10 ** (ebno_db / 10.0)) / 2.0)       # |x|^2 = 4, Es = 4 Ts, Eb = Es/2
        xn = x[:nsamp + 24] + rng.normal(0.0, sigma, (nsamp + 24, 2)).astype(np.float32)
        u8 = np.clip(np.rint(127.0 + 14.0 * xn.astype(np.float64)), 0, 255).astype(np.uint8)
        d = torch.from_numpy(u8).cuda()
        for c in range(24):
            dev[c::24] = d[c:c + nsamp].unsqueeze(0)
        h4.reset(); ld.reset()
        dem(); dec(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_d = t_l = 0.0
        for _ in range(args.iters):
            h4.reset(); ld.reset()
            ev[0].record(st); dem(); ev[1].record(st); dec(); ev[2].record(st); torch.cuda.synchronize()
            t_d += ev[0].elapsed_time(ev[1]); t_l += ev[1].elapsed_time(ev[2])
        t_d /= args.iters; t_l /= args.iters
        # the product's path: one call, bit LLRs handed over on the device (pirip_hip_fsk_ldpc_rx_batch)
        def chain():
            ld.chain_batch(h4, dev.data_ptr(), nsamp * 2, nsamp, stt.data_ptr(), pay.data_ptr(), inf.data_ptr(), nfr.data_ptr(), cons.data_ptr(), maxf,
                           stream=st.cuda_stream)
        h4.reset(); ld.reset(); chain(); torch.cuda.synchronize()
        assert ld.last_path_fused()
        t_c = 0.0
        for _ in range(args.iters):
            h4.reset(); ld.reset()
            ev[0].record(st); chain(); ev[1].record(st); torch.cuda.synchronize()
            t_c += ev[0].elapsed_time(ev[1])
        t_c /= args.iters
        # the chain once more with both handles' demodulator on the opt-in band-only estimator: time, and the records against the above
        band_chain = None
        try:
            ref_rec = (stt.clone(), pay.clone(), inf[..., 4:9].clone())
            h4.set_estimator_band_only(True)
            h4.reset(); ld.reset(); chain(); torch.cuda.synchronize()
            same = bool(torch.equal(stt, ref_rec[0]) and torch.equal(pay, ref_rec[1]) and torch.equal(inf[..., 4:9], ref_rec[2]))
            t_b = 0.0
            for _ in range(args.iters):
                h4.reset(); ld.reset()
                ev[0].record(st); chain(); ev[1].record(st); torch.cuda.synchronize()
                t_b += ev[0].elapsed_time(ev[1])
            t_b /= args.iters
            band_chain = {"chain_ms": t_b, "Msamples_per_s_end_to_end": float(cons.sum()) / t_b / 1e3, "records_identical_to_the_full_estimator": same}
            h4.set_estimator_band_only(False)
            h4.reset(); ld.reset(); chain(); torch.cuda.synchronize()       # (the entry's own figures below come from the full estimator's run)
        except Exception as e:
            band_chain = f"unavailable: {e!r}"
        okm = (stt & 4) != 0
        good = int(okm.sum())
        dec_frames = int((inf[..., 6] >= 0).sum())
        it = inf[..., 4][inf[..., 6] >= 0].float()
        eraw = inf[..., 8][okm].float()
        nsmp = float(cons.sum())
        e2e4 = nsmp / t_c / 1e3
        res[key] = {"workload": "BASELINE configs[3], whole chain: 4-FSK Fs=240k Rs=10k P=8 demod (soft decisions) -> FSK_LDPC receive, stand-in (512,256) code, Eb/N0 %.1f dB" % ebno_db,
                    "streams": B, "samples_per_stream": nsamp, "chain_ms": t_c,
                    "unfused_for_comparison": {"demod_ms_soft_magnitudes_out": t_d, "ldpc_rx_batch_ms": t_l},
                    "Msamples_per_s_end_to_end": e2e4, "frames_decoded": dec_frames, "frames_ok": good, "frames_ok_per_s": good / (t_c * 1e-3),
                    "opt_in_band_only_estimator": band_chain,
                    "mean_iterations": float(it.mean()) if dec_frames else None,
                    "raw_ber_of_delivered_frames": float(eraw.mean()) / 512.0 if good else None,
                    "roofline": {"bound": "hbm", "achieved": e2e4 * 1e6 * ab4 / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": e2e4 * 1e6 * ab4 / 1e9 / 8000.0, "algorithmic_bytes_per_sample": ab4,
                                 "designed_intermediate_bytes_per_sample": inter4}}
    del dev, filt, stt, pay, inf, h4, ld

    # ---- config 3: 64 streams x 45e6 samples at 1.8 MS/s -> /45 -> demod at 40 kS/s ----------
    B, n_lo = int(os.environ.get("PIRIP_CFG3_STREAMS", "4096")), 40_000
    x, _ = modulate(L, 40000, 1000, 2, 1000, 2000, n_lo // 40 + 50, 2)
    xl = torch.from_numpy(x[:n_lo + 2]).cuda()
    t = torch.arange((n_lo) * 45, device="cuda", dtype=torch.float64) / 45.0
    i0 = t.floor().long(); fr = (t - i0).float().unsqueeze(1)
    hi = (1 - fr) * xl[i0] + fr * xl[i0 + 1]                        # x45 linear interpolation (tlininterp)
    u8 = torch.clamp(torch.round(127.0 + 40.0 * hi.double()), 0, 255).to(torch.uint8)
    n_in = u8.shape[0]
    dev = u8.unsqueeze(0).expand(B, n_in, 2).contiguous()
    del t, i0, fr, hi
    dec = pirip_amd.HipDecim(45, 0.05, out_s16=True)
    n_out = dec.nout(n_in)
    mid = torch.zeros((B, n_out, 2), dtype=torch.int16, device="cuda")
    h3 = pirip_amd.HipDemod(40000, 1000, 2, P=8, est_min=500, est_max=20000, in_format=pirip_amd.IN_CS16, nstreams=B)
    maxf = h3.max_frames_for(n_out)
    bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")

    def run3():
        dec.batch(dev.data_ptr(), n_in * 2, n_in, mid.data_ptr(), n_out * 4, B, st.cuda_stream)
        h3.demod_batch(mid.data_ptr(), n_out * 4, n_out, bits.data_ptr(), maxf * 50, 0, 0, 0, 0,
                       nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
    run3(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tot_d = tot_m = 0.0
    for _ in range(args.iters):
        ev[0].record(st)
        dec.batch(dev.data_ptr(), n_in * 2, n_in, mid.data_ptr(), n_out * 4, B, st.cuda_stream)
        ev[1].record(st)
        h3.demod_batch(mid.data_ptr(), n_out * 4, n_out, bits.data_ptr(), maxf * 50, 0, 0, 0, 0,
                       nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
        ev[2].record(st); torch.cuda.synchronize()
        tot_d += ev[0].elapsed_time(ev[1]); tot_m += ev[1].elapsed_time(ev[2])
    md, mm = tot_d / args.iters, tot_m / args.iters
    e2e = B * n_in / (md + mm) / 1e3
    # algorithmic bytes per 1.8 MS/s input sample: 2 B read by the decimator + one byte per bit out (50 / 90000 B);
    # the decimator's s16 output written and read back by the demodulator (2 x 4/45 B) is intermediate traffic, reported beside it
    ab = 2.0 + 50.0 / 90000.0
    inter3 = 8.0 / 45.0
    res["config3_decim45_then_demod"] = {"workload": "BASELINE configs[2]: u8 IQ at 1.8 MS/s -> csdr /45 decimator (s16 out) -> 2-FSK Rs=1k demod at 40 kS/s",
                                         "streams": B, "input_samples_per_stream": n_in, "decim_ms": md, "demod_ms": mm,
                                         "input_Msamples_per_s_end_to_end": e2e,
                                         "demod_Msamples_per_s_at_40k": B * n_out / mm / 1e3,
                                         "frames_per_stream": int(nfr[0]),
                                         "roofline": {"bound": "hbm", "achieved": e2e * 1e6 * ab / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                      "frac": e2e * 1e6 * ab / 1e9 / 8000.0, "algorithmic_bytes_per_input_sample": ab,
                                                      "designed_intermediate_bytes_per_input_sample": inter3,
                                                      "decimator_alone_frac": B * n_in * (2.0 + 4.0 / 45.0) / (md * 1e-3) / 1e9 / 8000.0}}
    # The decimator's ordering contract, with numbers: the two OPT-IN tap-loop arithmetics against the default
    # (exact = the scalar csdr loop bit for bit): rate, s16 outputs that differ, whether config 3's decoded bits stay the same
    h3.reset(st.cuda_stream)
    run3(); torch.cuda.synchronize()                                   # the exact path once from the reset state: what the variants are compared with
    mid0, bits0, nfr0 = mid.clone(), bits.clone(), nfr.clone()
    opt = {}
    for mode, name in ((1, "fma"), (2, "fma_raw")):
        dec.set_arith(mode)
        h3.reset(st.cuda_stream)
        run3(); torch.cuda.synchronize()
        diff = (mid.int() - mid0.int()).abs()
        same_bits = bool(torch.equal(nfr, nfr0) and torch.equal(bits, bits0))
        td = tm = 0.0
        for _ in range(args.iters):
            ev[0].record(st)
            dec.batch(dev.data_ptr(), n_in * 2, n_in, mid.data_ptr(), n_out * 4, B, st.cuda_stream)
            ev[1].record(st)
            h3.demod_batch(mid.data_ptr(), n_out * 4, n_out, bits.data_ptr(), maxf * 50, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
            ev[2].record(st); torch.cuda.synchronize()
            td += ev[0].elapsed_time(ev[1]); tm += ev[1].elapsed_time(ev[2])
        td, tm = td / args.iters, tm / args.iters
        e2 = B * n_in / (td + tm) / 1e3
        opt[name] = {"decim_ms": td, "demod_ms": tm, "input_Msamples_per_s_end_to_end": e2, "frac": e2 * 1e6 * ab / 1e9 / 8000.0,
                     "decimator_alone_frac": B * n_in * (2.0 + 4.0 / 45.0) / (td * 1e-3) / 1e9 / 8000.0,
                     "s16_outputs_compared": int(mid0.numel()), "s16_outputs_differing": int((diff > 0).sum()), "largest_difference_lsb": int(diff.max()),
                     "decoded_bits_equal_to_exact_path": same_bits}
    dec.set_arith(0)
    res["config3_decim45_then_demod"]["opt_in_tap_arithmetic"] = dict(opt, note="default stays exact (the scalar csdr loop's float32 result bit for bit, "
        "oracle/csdr_oracle.c); fma = accumulation fused; fma_raw = the u8 affine map pulled out of the sum (half the VALU instructions); "
        "h3 state reset before each variant's comparison run")
    del dev, mid, bits, mid0, bits0

    # ---- rtl_fsk's in-process decimations (u8 at the RTL rate -> complex float at the modem rate): -a 40000 at 240 kS/s (script/ping:47,
    #      script/frame_repeater:36) = /6, -a 100000 at 1.8 MS/s (README.md:196) = /18 -- the systolic decimator kernel
    sysd = {}
    for D in (6, 18):
        Bd, n_in_d = 64, D * 1_000_000
        devd = torch.randint(0, 256, (Bd, n_in_d, 2), dtype=torch.uint8, device="cuda")
        decd = pirip_amd.HipDecim(D, 0.05, out_s16=False)
        n_out_d = decd.nout(n_in_d)
        outd = torch.zeros((Bd, n_out_d, 2), dtype=torch.float32, device="cuda")
        rund = lambda: decd.batch(devd.data_ptr(), n_in_d * 2, n_in_d, outd.data_ptr(), n_out_d * 8, Bd, st.cuda_stream)
        rund(); torch.cuda.synchronize()
        e0.record(st)
        for _ in range(args.iters):
            rund()
        e1.record(st); torch.cuda.synchronize()
        msd = e0.elapsed_time(e1) / args.iters
        abd = 2.0 + 8.0 / D
        sysd[f"div{D}"] = {"streams": Bd, "input_samples_per_stream": n_in_d, "kernel_ms": msd, "input_Msamples_per_s": Bd * n_in_d / msd / 1e3,
                           "frac_of_hbm_roofline": Bd * n_in_d * abd / (msd * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_input_sample": abd}
        del devd, outd, decd
    res["rtl_fsk_inprocess_decimators"] = dict(sysd, workload="rtl_fsk -a <modem rate>: u8 IQ at the RTL rate -> csdr windowed-sinc decimator (79 taps) -> complex float, device-resident; bit-exact")

    # ---- rtl_fsk -r 1000 at 240 kS/s (README.md:152,184 / :239): Ts = 240, Ndft = 4096 on the workgroup-per-stream instance -------
    for M, mask, key in ((2, 0, "rtl_fsk_r1000_2fsk_block"), (4, 2000, "rtl_fsk_r1000_4fsk_mask_block")):
        B, nsamp = 6144, 24 * 12000
        x, _ = modulate(L, 240000, 1000, M, 11000, 2000, nsamp // 240 + 50, 7)
        x = x[:nsamp] + 0.2 * np.random.default_rng(3).standard_normal((nsamp, 2)).astype(np.float32)
        u8 = np.clip(np.rint(127.5 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
        dev = torch.from_numpy(u8).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
        hb = pirip_amd.HipDemod(240000, 1000, M, P=15, est_min=500, est_max=90000, mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=B)
        maxf = hb.max_frames_for(nsamp)
        nb = 50 * (1 if M == 2 else 2)
        bits = torch.zeros((B, maxf, nb), dtype=torch.uint8, device="cuda")
        nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
        runb = lambda: hb.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * nb, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
        runb(); torch.cuda.synchronize()
        e0.record(st)
        for _ in range(args.iters):
            runb()
        e1.record(st); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        r = float(cons.sum()) / ms / 1e3
        ab = 2.0 + nb / 12000.0
        res[key] = {"workload": f"rtl_fsk -r 1000 at 240 kS/s{' -m 4 --mask 2000' if M == 4 else ''} (README.md:{'239' if M == 4 else '152,184'}): Ts=240 P=15 Ndft=4096, u8 IQ (csdr mapping), device-resident",
                    "kernel": hb.kernel(), "streams": B, "samples_per_stream": nsamp, "kernel_ms": ms, "Msamples_per_s": r,
                    "roofline": {"bound": "hbm", "achieved": r * 1e6 * ab / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": r * 1e6 * ab / 1e9 / 8000.0,
                                 "algorithmic_bytes_per_sample": ab}}
        del dev, bits, hb

    # ---- rtl_fsk -a 40000 -r 1000 (the services' modem, script/ping:47, script/frame_repeater:36): the in-process decimator hands the demodulator
    # complex floats; Ts = 40, Ndft = 512 wave instance that reads its frames from global memory --------------------------
    B, nsamp = 3072, 100 * 2000
    x, _ = modulate(L, 40000, 1000, 2, 1000, 2000, nsamp // 40 + 50, 11)
    x = (x[:nsamp] + 0.2 * np.random.default_rng(5).standard_normal((nsamp, 2))).astype(np.float32)
    dev = torch.from_numpy(x).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
    hf = pirip_amd.HipDemod(40000, 1000, 2, P=10, est_min=500, est_max=19000, in_format=pirip_amd.IN_CF32, nstreams=B)
    maxf = hf.max_frames_for(nsamp)
    bits = torch.zeros((B, maxf, 50), dtype=torch.uint8, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    runf = lambda: hf.demod_batch(dev.data_ptr(), nsamp * 8, nsamp, bits.data_ptr(), maxf * 50, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
    runf(); torch.cuda.synchronize()
    e0.record(st)
    for _ in range(args.iters):
        runf()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    r = float(cons.sum()) / ms / 1e3
    ab = 8.0 + 50.0 / 2000.0
    res["rtl_fsk_a40000_r1000_f32_demod"] = {"workload": "demodulator side of rtl_fsk -a 40000 -r 1000 (script/ping:47, script/frame_repeater:36): 2-FSK Ts=40 P=10 Ndft=512, complex-float input, device-resident",
                                             "kernel": hf.kernel_name(), "streams": B, "samples_per_stream": nsamp, "kernel_ms": ms, "Msamples_per_s": r,
                                             "roofline": {"bound": "hbm", "achieved": r * 1e6 * ab / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": r * 1e6 * ab / 1e9 / 8000.0,
                                                          "algorithmic_bytes_per_sample": ab}}
    del dev, bits, hf
    return res


def chain_only(ebno_db, iters):
    """Profiling aid (tools/chain_traffic.sh): nothing but the config-4 whole chain (pirip_hip_fsk_ldpc_rx_batch) at one Eb/N0."""
    import subprocess
    import torch
    import pirip_amd
    L = pirip_amd.lib()
    st = torch.cuda.current_stream()
    B, nsamp = 8192, 600_000
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = subprocess.run([framer, "--code", pirip_amd.STANDIN_CODE, "--testframes", "93", "--seq", "--source", "0x1", "/dev/zero", "-"],
                        capture_output=True, check=True).stdout
    x, _ = modulate(L, 240000, 10000, 4, 10000, 10000, 0, 3, bits=np.frombuffer(fb, dtype=np.uint8))
    rng = np.random.default_rng(5)
    sigma = np.sqrt((4.0 * 24 / 2.0) / (10 ** (ebno_db / 10.0)) / 2.0)
    xn = x[:nsamp + 24] + rng.normal(0.0, sigma, (nsamp + 24, 2)).astype(np.float32)
    u8 = np.clip(np.rint(127.0 + 14.0 * xn.astype(np.float64)), 0, 255).astype(np.uint8)
    d = torch.from_numpy(u8).cuda()
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    for c in range(24):
        dev[c::24] = d[c:c + nsamp].unsqueeze(0)
    h4 = pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=B)
    maxf = h4.max_frames_for(nsamp)
    ld = pirip_amd.HipLdpc(pirip_amd.STANDIN_CODE, 4, nstreams=B)
    stt = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda")
    pay = torch.zeros((B, maxf, 32), dtype=torch.uint8, device="cuda")
    inf = torch.zeros((B, maxf, pirip_amd.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    for _ in range(iters + 1):
        h4.reset(); ld.reset()
        ld.chain_batch(h4, dev.data_ptr(), nsamp * 2, nsamp, stt.data_ptr(), pay.data_ptr(), inf.data_ptr(), nfr.data_ptr(), cons.data_ptr(), maxf,
                       stream=st.cuda_stream)
        torch.cuda.synchronize()
    return {"ebno_db": ebno_db, "fused": ld.last_path_fused(), "samples_per_call": float(cons.sum()), "frames_ok": int(((stt & 4) != 0).sum())}


def main():
    if os.environ.get("PIRIP_CHAIN_ONLY"):
        print(json.dumps(chain_only(float(os.environ["PIRIP_CHAIN_ONLY"]), 2)))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    print(json.dumps(measure(args.iters)))


if __name__ == "__main__":
    main()
