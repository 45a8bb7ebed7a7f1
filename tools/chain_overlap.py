#!/usr/bin/env python3
"""tools/chain_overlap.py [ebno ...] -- BASELINE configs[3]'s whole chain with the batch cut into K groups of streams, each group its own
pair of handles (HipDemod + HipLdpc) on its own HIP stream, earlier groups at higher stream priority: group k's LDS-bound LDPC stages
then run beside group k+1's VALU-bound demodulator instead of after the whole batch's. Same streams, same records (checksum printed);
K = 1 is the single call `bench.py` / `tools/bench_configs.py` time. What a caller can do today with the C-ABI as it stands."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import numpy as np
    import torch
    import pirip_amd
    import bench_configs
    L = pirip_amd.lib()
    B, nsamp = int(os.environ.get("AB_STREAMS", "8192")), 600_000
    framer = os.path.join(ROOT, "pirip_amd", "bin", "fsk_ldpc_framer")
    fb = subprocess.run([framer, "--code", pirip_amd.STANDIN_CODE, "--testframes", "93", "--seq", "--source", "0x1", "/dev/zero", "-"],
                        capture_output=True, check=True).stdout
    x, _ = bench_configs.modulate(L, 240000, 10000, 4, 10000, 10000, 0, 3, bits=np.frombuffer(fb, dtype=np.uint8))
    dev = torch.empty((B, nsamp, 2), dtype=torch.uint8, device="cuda")
    it = int(os.environ.get("AB_ITERS", "8"))
    ebnos = [float(v) for v in sys.argv[1:]] or [7.0, 3.5]
    print(f"# config-4 chain, {B} streams x {nsamp} samples, cut into K groups on K prioritised HIP streams; ms per batch, G samples/s, checksum of all records")
    for ebno_db in ebnos:
        rng = np.random.default_rng(5)
        sigma = np.sqrt((4.0 * 24 / 2.0) / (10 ** (ebno_db / 10.0)) / 2.0)
        xn = x[:nsamp + 24] + rng.normal(0.0, sigma, (nsamp + 24, 2)).astype(np.float32)
        u8 = np.clip(np.rint(127.0 + 14.0 * xn.astype(np.float64)), 0, 255).astype(np.uint8)
        d = torch.from_numpy(u8).cuda()
        for c in range(24):
            dev[c::24] = d[c:c + nsamp].unsqueeze(0)
        splits = [[B], [B // 2, B // 2], [B * 5 // 8, B * 3 // 8], [B * 3 // 4, B // 4], [B // 2, B * 5 // 16, B * 3 // 16], [B // 4] * 4]
        if os.environ.get("AB_SPLIT"):
            splits = [[int(v) for v in sp.split(",")] for sp in os.environ["AB_SPLIT"].split(";")]
        for groups in splits:
            K = len(groups)
            offs = [sum(groups[:k]) for k in range(K)]
            hs = [pirip_amd.HipDemod(240000, 10000, 4, P=8, est_min=500, est_max=60000, nstreams=g) for g in groups]
            lds = [pirip_amd.HipLdpc(pirip_amd.STANDIN_CODE, 4, nstreams=g) for g in groups]
            maxf = hs[0].max_frames_for(nsamp)
            stt = torch.zeros((B, maxf), dtype=torch.uint8, device="cuda")
            pay = torch.zeros((B, maxf, 32), dtype=torch.uint8, device="cuda")
            inf = torch.zeros((B, maxf, pirip_amd.LDPC_INFO_PER_CALL), dtype=torch.int32, device="cuda")
            nfr = torch.zeros(B, dtype=torch.int32, device="cuda")
            cons = torch.zeros(B, dtype=torch.int64, device="cuda")
            sts = [torch.cuda.Stream(priority=-1 if (k < K - 1 and K > 1) else 0) for k in range(K)]
            main_st = torch.cuda.current_stream()

            def run():
                e = torch.cuda.Event(); e.record(main_st)
                for k in range(K):
                    sts[k].wait_event(e)
                    a = offs[k]
                    lds[k].chain_batch(hs[k], dev[a].data_ptr(), nsamp * 2, nsamp, stt[a].data_ptr(), pay[a].data_ptr(), inf[a].data_ptr(), nfr[a].data_ptr(),
                                       cons[a].data_ptr(), maxf, stream=sts[k].cuda_stream)
                for k in range(K):
                    e2 = torch.cuda.Event(); e2.record(sts[k]); main_st.wait_event(e2)
            for _ in range(2):
                for k in range(K):
                    hs[k].reset(); lds[k].reset()
                run()
            torch.cuda.synchronize()
            t = 0.0
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(it):
                for k in range(K):
                    hs[k].reset(); lds[k].reset()
                torch.cuda.synchronize()
                ev[0].record(main_st); run(); ev[1].record(main_st); torch.cuda.synchronize()
                t += ev[0].elapsed_time(ev[1])
            t /= it
            hsh = hashlib.sha256()
            for tt in (stt, pay, inf[..., 4:9].contiguous()):
                hsh.update(tt.cpu().numpy().tobytes())
            print(f"{ebno_db:4.1f} dB  groups {groups}: {t:6.2f} ms  {float(cons.sum()) / t / 1e6:6.1f} G  ok {int(((stt & 4) != 0).sum())}  rec {hsh.hexdigest()[:12]}", flush=True)
            del hs, lds, stt, pay, inf


if __name__ == "__main__":
    main()
