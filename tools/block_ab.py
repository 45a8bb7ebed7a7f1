import os, sys, json, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np, torch, pirip_amd, bench_configs
    M = int(sys.argv[2]); mask = 2000 if M == 4 else 0
    B, nsamp = 6144, 24 * 12000
    x, _ = bench_configs.modulate(pirip_amd.lib(), 240000, 1000, M, 11000, 2000, nsamp // 240 + 50, 7)
    x = x[:nsamp] + 0.2 * np.random.default_rng(3).standard_normal((nsamp, 2)).astype(np.float32)
    u8 = np.clip(np.rint(127.5 + 32.0 * x.astype(np.float64)), 0, 255).astype(np.uint8)
    dev = torch.from_numpy(u8).cuda().unsqueeze(0).expand(B, nsamp, 2).contiguous()
    hb = pirip_amd.HipDemod(240000, 1000, M, P=15, est_min=500, est_max=90000, mask=mask, in_format=pirip_amd.IN_CU8_CSDR, nstreams=B)
    maxf = hb.max_frames_for(nsamp); nb = 50 * (1 if M == 2 else 2)
    bits = torch.zeros((B, maxf, nb), dtype=torch.uint8, device="cuda"); nfr = torch.zeros(B, dtype=torch.int32, device="cuda"); cons = torch.zeros(B, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream()
    run = lambda: hb.demod_batch(dev.data_ptr(), nsamp * 2, nsamp, bits.data_ptr(), maxf * nb, 0, 0, 0, 0, nfr.data_ptr(), cons.data_ptr(), maxf, st.cuda_stream)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): run()
    e1.record(st); torch.cuda.synchronize()
    print("RATE", float(cons.sum()) / (e0.elapsed_time(e1) / 20) / 1e6, hb.kernel())
else:
    # variants: "TAG=ENVNAME=VALUE" (environment of that variant's processes) on the command line, default: the start-up stagger experiment
    variants = [v.split("=", 1) for v in sys.argv[1:]] or [["A default", ""], ["B stagger 3", "PIRIP_BLOCK_STAGGER=3"], ["C stagger 6", "PIRIP_BLOCK_STAGGER=6"]]
    for M in (2, 4):
        for rep in range(3):
            for tag, envs in variants:
                env = dict(os.environ)
                env.update(kv.split("=", 1) for kv in envs.split(",") if "=" in kv)
                r = subprocess.run([sys.executable, __file__, "child", str(M)], env=env, capture_output=True, text=True)
                print(M, tag, [l for l in r.stdout.splitlines() if l.startswith("RATE")] or r.stderr[-300:], flush=True)
