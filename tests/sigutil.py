"""Synthetic IQ generation shared by the tests (uses the CPU oracle's modulator).

Configurations follow BASELINE.json / SURVEY.md 8d:
  cfg1/2: 2-FSK Fs=240k Rs=10k P=24, tones 10/20 kHz, u8 IQ, `fsk_demod -d -p 24`
  cfg3  : u8 IQ at 1.8 MS/s -> /45 -> 2-FSK Rs=1k at 40 kS/s, `fsk_demod -c 2 40000 1000`
  cfg4  : 4-FSK Fs=240k Rs=10k, tones 10/20/30/40 kHz
"""
import numpy as np

CFG1 = dict(Fs=240000, Rs=10000, M=2, P=24, f1=10000, shift=10000, est_min=500, est_max=25000)
CFG3 = dict(Fs=40000, Rs=1000, M=2, P=8, f1=1000, shift=2000, est_min=500, est_max=20000)
CFG4 = dict(Fs=240000, Rs=10000, M=4, P=8, f1=10000, shift=10000, est_min=500, est_max=60000)


def mod_complex(ob, cfg, bits, f1=None):
    tx = ob.OracleFsk(cfg["Fs"], cfg["Rs"], cfg["M"], P=cfg["P"], f1_tx=f1 if f1 is not None else cfg["f1"],
                      tone_spacing=cfg["shift"])
    return tx.mod_c(bits)     # float32 [n,2], peak 2.0


def add_awgn(x, ebno_db, cfg, rng):
    """complex AWGN for a given Eb/N0 (signal power = |x|^2 mean, Eb = Ps*Ts/log2(M))."""
    ps = float(np.mean(x[:, 0].astype(np.float64) ** 2 + x[:, 1].astype(np.float64) ** 2))
    ts = cfg["Fs"] // cfg["Rs"]
    eb = ps * ts / np.log2(cfg["M"])
    n0 = eb / (10 ** (ebno_db / 10.0))
    sigma = np.sqrt(n0 / 2.0)
    return (x + rng.normal(0.0, sigma, x.shape)).astype(np.float32)


def make_u8_stream(ob, cfg, nbits, seed=0, offset=0, tone_bins=0, ebno_db=None, random_bits=False, amp=32.0,
                   noise_amp_scale=1.0):
    """u8 IQ [n,2] for one stream. offset: leading samples dropped (timing phase);
    tone_bins: tone plan shifted by k*Fs/Ndft-ish (k*937.5 Hz at cfg1)."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, nbits).astype(np.uint8) if random_bits else ob.get_test_bits(nbits)
    f1 = cfg["f1"] + int(round(tone_bins * 937.5))
    x = mod_complex(ob, cfg, bits, f1=f1)
    if ebno_db is not None:
        x = add_awgn(x, ebno_db, cfg, rng)
        amp = amp * noise_amp_scale
    u8 = ob.quantise_cu8(x, amp=amp)
    return np.ascontiguousarray(u8[offset:]), bits


def rel_err(a, b):
    """max |a-b| relative to the peak of b, per call (the stated rx_filt tolerance metric)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    peak = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / peak


def code_variant(src, outdir, llr_map):
    """A copy of code file `src` with the `llr_map` key set (pirip_amd/csrc/fsk_ldpc.hpp: upstream | rician), for tests that run both
    soft-decision mappings of the FSK_LDPC receiver. Returns the new path."""
    import os
    lines = [ln for ln in open(src).read().split("\n") if not ln.startswith("llr_map")]
    i = next(k for k, ln in enumerate(lines) if ln.startswith("rows"))
    lines.insert(i, "llr_map " + llr_map)
    out = os.path.join(str(outdir), os.path.basename(src).replace(".code", "_" + llr_map + ".code"))
    with open(out, "w") as f:
        f.write("\n".join(lines))
    return out
