#!/usr/bin/env python3
"""tools/chain_ab.py LIB_A LIB_B [LIB_C ...] [ebno ...] -- A/B comparison of builds of libpirip_hip.so on BASELINE configs[3]'s whole chain
(pirip_hip_fsk_ldpc_rx_batch: 4-FSK demodulator with the fused hand-over -> unique-word search -> sync -> LDPC decode -> CRC16) and on
the stand-alone FSK_LDPC receive stage fed with soft magnitudes. The two libraries are loaded in separate processes alternately
(A B A B A B): clocks and power state drift by a few per cent between runs, so only interleaved repeats are comparable. Every run
also prints a checksum of the records (status, payload, iteration counts) so that a change of results cannot hide in a rate."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import torch
    import pirip_amd
    import bench_configs
    L = pirip_amd.lib()
    st = torch.cuda.current_stream()
    B, nsamp = int(os.environ.get("AB_STREAMS", "8192")), 600_000
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = subprocess.run([framer, "--code", pirip_amd.STANDIN_CODE, "--testframes", "93", "--seq", "--source", "0x1", "/dev/zero", "-"],
                        capture_output=True, check=True).stdout
    x, _ = bench_configs.modulate(L, 240000, 10000, 4, 10000, 10000, 0, 3, bits=np.frombuffer(fb, dtype=np.uint8))
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    h4 = pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=B)
    maxf = h4.max_frames_for(nsamp)
    ld = pirip_amd.HipLdpc(pirip_amd.STANDIN_CODE, 4, nstreams=B)
    stt = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda")
    pay = torch.zeros((B, maxf, 32), dtype=torch.uint8, device="cuda")
    inf = torch.zeros((B, maxf, pirip_amd.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")
    nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
    cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    filt = torch.zeros((B, maxf, 200), dtype=torch.float32, device="cuda")
    out = {}
    for ebno_db in (float(v) for v in sys.argv[2].split(",")):
        rng = np.random.default_rng(5)
        sigma = np.sqrt((4.0 * 24 / 2.0) / (10 ** (ebno_db / 10.0)) / 2.0)
        xn = x[:nsamp + 24] + rng.normal(0.0, sigma, (nsamp + 24, 2)).astype(np.float32)
        u8 = np.clip(np.rint(127.0 + 14.0 * xn.astype(np.float64)), 0, 255).astype(np.uint8)
        d = torch.from_numpy(u8).cuda()
        for c in range(24):
            dev[c::24] = d[c:c + nsamp].unsqueeze(0)

        def chain():
            ld.chain_batch(h4, dev.data_ptr(), nsamp * 2, nsamp, stt.data_ptr(), pay.data_ptr(), inf.data_ptr(), nfr.data_ptr(), cons.data_ptr(), maxf,
                           stream=st.cuda_stream)

        def dem():
            h4.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, 0, 0, filt.data_ptr(), maxf * 200, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)

        def dec():
            ld.rx_batch(filt.data_ptr(), maxf * 200, nfr.data_ptr(), maxf, stt.data_ptr(), pay.data_ptr(), inf.data_ptr(), st.cuda_stream)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        it = int(os.environ.get("AB_ITERS", "8"))
        for _ in range(2):
            h4.reset(); ld.reset(); chain()
        torch.cuda.synchronize()
        t_c = 0.0
        for _ in range(it):
            h4.reset(); ld.reset()
            ev[0].record(st); chain(); ev[1].record(st); torch.cuda.synchronize()
            t_c += ev[0].elapsed_time(ev[1])
        hsh = hashlib.sha256()
        for t in (stt, pay, inf[..., 4:9].contiguous()):
            hsh.update(t.cpu().numpy().tobytes())
        t_l = 0.0
        for _ in range(it):
            h4.reset(); ld.reset(); dem()
            ev[0].record(st); dec(); ev[1].record(st); torch.cuda.synchronize()
            t_l += ev[0].elapsed_time(ev[1])
        valid = inf[..., 6] >= 0
        out[ebno_db] = {"chain_ms": t_c / it, "ldpc_rx_batch_ms": t_l / it, "G": float(cons.sum()) / (t_c / it) / 1e6,
                        "frames_ok": int(((stt & 4) != 0).sum()), "mean_it": float(inf[..., 4][valid].float().mean()), "records": hsh.hexdigest()[:12]}
    print("ABCHAIN " + json.dumps(out))


def main():
    args = sys.argv[1:]
    # a variant is LIB.so or LIB.so:NAME=VALUE[,NAME=VALUE...] (environment of that variant's processes, e.g. PIRIP_LDPC_DECODER=fast)
    libs = [a for a in args if ".so" in a]
    ebnos = ",".join([a for a in args if ".so" not in a] or ["7", "3.5"])
    print(f"## config-4 chain, Eb/N0 {ebnos} dB: runs interleaved A B ... A B ... A B ...")
    for rep in range(3):
        for tag, lib in zip("ABCDEFGH", libs):
            lib, _, extra = lib.partition(":")
            env = dict(os.environ, PIRIP_HIP_LIB=os.path.abspath(lib))
            env.update(kv.split("=", 1) for kv in extra.split(",") if "=" in kv)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ebnos], env=env, capture_output=True, text=True)
            ln = [l for l in r.stdout.splitlines() if l.startswith("ABCHAIN ")]
            if not ln:
                print(tag, "failed", r.stderr[-600:]); continue
            d = json.loads(ln[0][8:])
            print(f"{tag} {os.path.basename(os.path.dirname(lib)) + (' ' + extra if extra else ''):8s} " + "   ".join(
                f"{e} dB: chain {v['chain_ms']:6.2f} ms ({v['G']:5.1f} G) rx_batch {v['ldpc_rx_batch_ms']:5.2f} ms ok {v['frames_ok']} it {v['mean_it']:.2f} rec {v['records']}"
                for e, v in d.items()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
