#!/usr/bin/env python3
"""tools/valu_floor.py -- how many VALU wave instructions per frame does the demodulator's arithmetic NEED?

The wave kernels are VALU-issue bound, so "fraction of the HBM roofline" says little about how good they are;
this tool gives the other yardstick: for a shape (M, Ts, P, Nsym, Ndft, input format) it walks the loop bounds of the CPU
restatement (oracle/fsk_oracle.c, line numbers cited per phase) and counts

  oracle_flops   the float operations the scalar C executes per frame, as written (O(P)-redundant window sums included);
  floor_ops      the lane operations (mul / add / fma / conversion / compare ...; an fma counts once) an implementation MUST execute under this repo's parity contract:
                   * estimator (a-5): Sf / f_est are bit-exact, so every multiply and add of the window, the kiss_fft butterflies,
                     |X|^2, the correctly rounded sqrt and the IIR has to be performed with the same operands in the same order;
                     only operations that are exact identities may go (twiddle (1,-0) multiplies of the k = 0 butterflies and of
                     the radix-2 leaves; the fftshift, which is index arithmetic);
                   * correlator / timing / decisions (a-6..a-8): under a stated tolerance, so the cheapest known algebra counts:
                     one oscillator step + one mix per sample and tone, running prefix sums (1 complex add per sample and tone)
                     with one complex subtraction per window instead of Ts-term sums;
  floor_instr    floor_ops turned into wave64 VALU instructions at perfect lane use: one instruction covers 64 lanes; mul / add /
                 fma on float pairs pack two per lane (v_pk_*_f32); conversions, transcendentals, compares / selects and cross-lane
                 (DPP) steps do not pack. No address arithmetic, no LDS traffic, no loop control: a LOWER bound on issue slots.

`python tools/valu_floor.py` prints the table for the headline shape; `--json` emits {"phases": ..., "floor_instr_per_frame": ...}
(bench.py imports floor() to put valu.floor_instr_per_frame and executed / floor in its line); `--executed a,b,c,...` adds the
kernel's measured per-phase SQ_INSTS_VALU (tools/phase_valu.sh) beside the floor."""
import argparse
import json
import math

LANES = 64


def fft_factors(n):
    """kiss_fft's kf_factor: 4s first, then 2s (Ndft is a power of two here); listed root stage first."""
    f = []
    while n % 4 == 0:
        f.append(4); n //= 4
    while n % 2 == 0:
        f.append(2); n //= 2
    assert n == 1
    return f


def fft_ops(ndft):
    """(complex adds, complex multiplies actually needed, complex multiplies kiss_fft executes) of one forward FFT.
    Stage with radix p over sub-transforms of length m: Ndft/p butterflies; butterfly k of a group has twiddles tw^(k*fstride*{1,2,3}),
    all equal to (1,-0) when k = 0 -- Ndft/(p*m) such butterflies per stage. kf_bfly4: 3 C_MUL + 8 complex add/sub; kf_bfly2: 1 C_MUL + 2."""
    fac = fft_factors(ndft)
    cadd = cmul_need = cmul_exec = 0
    m = 1
    for p in reversed(fac):                      # leaf stage (m = 1) first
        nb = ndft // p
        trivial = ndft // (p * m)
        cadd += nb * (8 if p == 4 else 2)
        cmul_exec += nb * (3 if p == 4 else 1)
        cmul_need += (nb - trivial) * (3 if p == 4 else 1)
        m *= p
    return cadd, cmul_need, cmul_exec


def fft_ops_pruned(ndft, needed):
    """the same count when only the output bins in `needed` are wanted (the opt-in band-only estimator): walks
    kiss_fft's decimation-in-time recursion from the root stage down, keeping per stage the butterflies that feed a wanted output.
    A radix-4 butterfly with wanted outputs S of {0,1,2,3}: its three twiddle multiplies, 2 + |S & {0,2}| adds if S meets {0,2}
    (F0 + s1, s0 + s2, then one per output) and 2 + |S & {1,3}| if it meets {1,3}; a radix-2 butterfly 1 multiply + one add per output."""
    fac = fft_factors(ndft)

    def rec(n, need, level):
        if n == 1 or not need:
            return 0, 0
        p = fac[level]
        m = n // p
        cadd = cmul = 0
        sub = set()
        for k in range(m):
            outs = {j for j in range(p) if k + j * m in need}
            if not outs:
                continue
            sub.add(k)
            if p == 4:
                n02, n13 = len(outs & {0, 2}), len(outs & {1, 3})
                cadd += (2 + n02 if n02 else 0) + (2 + n13 if n13 else 0)
                cmul += 3 if k else 0
            else:
                cadd += len(outs)
                cmul += 1 if k else 0
        a, c = rec(m, sub, level + 1)
        return cadd + p * a, cmul + p * c

    return rec(ndft, set(needed), 0)


def floor(M=2, Ts=24, P=24, Nsym=50, Ndft=256, fmt="u8", band_bins=None):
    """band_bins: None = the full estimator (Sf of all Ndft bins, the default and what bench.py quotes); k = the opt-in band-only
    estimator keeping FFT bins 0 .. k-1"""
    N = Ts * Nsym
    nfft = N // (Ndft // 2) - 1                   # fsk_oracle.c:162 (nin = N)
    nint = (Nsym + 1) * P                        # fsk_oracle.c:292
    bins = Ndft * nfft
    ph = []

    def add(name, cite, oracle_flops, pk_ops, plain_ops, note=""):
        """pk_ops: lane-operations (mul / add / fma, an fma counting once) that pack two per instruction; plain_ops: lane-operations
        that do not pack"""
        ph.append({"phase": name, "oracle": cite, "oracle_flops": oracle_flops, "floor_ops": pk_ops + plain_ops,
                   "floor_instr": (pk_ops / 2.0 + plain_ops) / LANES, "note": note})

    # ---- a-1 conversion: each distinct sample once (the FFT windows overlap by half and the correlator reads the same samples)
    conv_plain, conv_pk = {"u8": (2, 2), "u8csdr": (2, 4), "s16": (2, 4), "f32": (0, 0)}[fmt]
    add("convert input (a-1)", "fsk_demod.c read loop; oracle_demod_buffer", N * (conv_plain + conv_pk), N * conv_pk, N * conv_plain,
        "2 integer->float conversions + the exact affine map per component, once per sample")
    # ---- a-5 estimator
    add("Hann window (a-5)", "fsk_oracle.c:165-169", bins * 2, bins * 2, 0, "2 multiplies per windowed sample, every FFT")
    cadd, cmul_need, cmul_exec = fft_ops(Ndft)
    cadd_o, cmul_o = cadd, cmul_exec
    if band_bins:
        cadd, cmul_need = fft_ops_pruned(Ndft, range(band_bins))
    add("kiss_fft butterflies (a-5)", "fsk_oracle.c:170; kiss_fft_oracle.c kf_bfly2/kf_bfly4", nfft * (cadd_o * 2 + cmul_o * 6),
        nfft * (cadd * 2 + cmul_need * 6), 0,
        f"{nfft} FFTs x ({cadd} complex add/sub + {cmul_need} non-trivial of {cmul_exec} complex multiplies, 4 mul + 2 add each, unfused)"
        + (f"; only what feeds bins 0..{band_bins - 1}" if band_bins else ""))
    bins_o = bins
    if band_bins:
        bins = band_bins * nfft
    add("|X|^2, sqrt, Sf IIR (a-5)", "fsk_oracle.c:180-188", bins_o * (3 + 1 + 3), bins * (3 + 4 + 3), bins * 2,
        "per bin and FFT: 2 mul + 1 add; a CORRECTLY ROUNDED sqrt = rsq + clamp (2 unpackable) + 4 mul/fma (v_sqrt_f32 alone is "
        "1 ulp off for 15 % of inputs, profiles/r02_sqrt_hw_error.txt); 2 mul + 1 add")
    span = band_bins or Ndft                      # est_min..est_max at most the whole spectrum (or the band)
    add("peak pick (a-5)", "fsk_oracle.c:195-211", M * span * 1, 0, M * (span * 2 + 2 * 6 * LANES),
        "per tone: compare + select per bin, a wave arg-max (6 max + 6 min cross-lane steps), blanking folded into the compare")
    # ---- a-6 down-conversion and integrator bank
    nin = N
    add("oscillators + mix (a-6)", "fsk_oracle.c:281-288", M * nin * 12, M * nin * 8, 0,
        "per sample and tone: phi *= dphi and in * conj(phi): 4 mul + 2 add each in the oracle, 2 mul + 2 fma each where fused "
        "multiply-add is allowed (tolerance rows)")
    add("window sums (a-6)", "fsk_oracle.c:293-301", M * nint * Ts * 2, M * (nin * 2 + nint * 2), 0,
        f"oracle: {nint} windows x {Ts} complex adds per tone; floor: running prefix sum (1 complex add per sample) + 1 complex "
        "subtraction per window")
    # ---- a-7 fine timing
    add("fine timing (a-7)", "fsk_oracle.c:304-337", nint * (M * 3 + (M - 1) + 4 + 6) + 60, nint * (M * 2 + (M - 1) + 2), 2 * 6 * LANES + 60 * LANES,
        "per window: |f_int|^2 per tone (mul + fma), sum over tones, x timing phasor (the recursion phi_ft *= dphift becomes a table: "
        "2 fma); two wave sums; atan2f + thresholds once per frame (~60 instructions on one lane = 60 wave instructions)")
    # ---- a-8 decisions
    add("resample + decide (a-8)", "fsk_oracle.c:340-385", Nsym * (M * 9 + M + 3), Nsym * M * 6, Nsym * (M * 2 + 2),
        "per symbol and tone: interpolation 2 mul + 2 fma, |t|^2 mul + fma; compare / select; (SNR sums: observable frames only, not counted)")
    tot = sum(p["floor_instr"] for p in ph)
    return {"shape": {"M": M, "Ts": Ts, "P": P, "Nsym": Nsym, "Ndft": Ndft, "fmt": fmt, "nfft": nfft, "nint": nint, "band_bins": band_bins},
            "phases": ph, "oracle_flops_per_frame": sum(p["oracle_flops"] for p in ph),
            "floor_ops_per_frame": sum(p["floor_ops"] for p in ph), "floor_instr_per_frame": tot,
            "floor_ops_per_sample": sum(p["floor_ops"] for p in ph) / N}


# how the kernel's timing marks (PIRIP_T_MARK, tools/phase_valu.sh) group the phases above
KERNEL_PHASES = [("estimator FFTs (window, FFT, |X|^2, sqrt, IIR; input conversion for the FFTs)", [0, 1, 2, 3]),
                 ("peak pick", [4]), ("correlator (conversion again, oscillators, mix, prefix sums)", [5]),
                 ("DMA issue, hist copy, window sums, timing reduction", [6, 7]), ("atan2, resample, decide, outputs", [8])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=2); ap.add_argument("--Ts", type=int, default=24); ap.add_argument("--P", type=int, default=24)
    ap.add_argument("--Nsym", type=int, default=50); ap.add_argument("--Ndft", type=int, default=256)
    ap.add_argument("--fmt", default="u8", choices=["u8", "u8csdr", "s16", "f32"])
    ap.add_argument("--band", type=int, default=None, help="the opt-in band-only estimator keeping FFT bins 0 .. BAND-1 (32 for the headline shape)")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--executed", default=None, help="comma list: the kernel's SQ_INSTS_VALU per frame in its five timing phases (tools/phase_valu.sh)")
    a = ap.parse_args()
    r = floor(a.M, a.Ts, a.P, a.Nsym, a.Ndft, a.fmt, a.band)
    if a.json:
        print(json.dumps(r)); return
    s = r["shape"]
    print(f"# VALU floor of one frame: M={s['M']} Ts={s['Ts']} P={s['P']} Nsym={s['Nsym']} Ndft={s['Ndft']} input {s['fmt']}: "
          f"{s['nfft']} FFTs, {s['nint']} integrator windows per tone, {s['Ts'] * s['Nsym']} samples")
    print(f"# {'phase':34s} {'oracle flops':>13s} {'floor ops':>12s} {'floor wave instr':>17s}   oracle lines")
    for p in r["phases"]:
        print(f"  {p['phase']:34s} {p['oracle_flops']:13d} {p['floor_ops']:12d} {p['floor_instr']:17.1f}   {p['oracle']}")
    print(f"  {'total':34s} {r['oracle_flops_per_frame']:13d} {r['floor_ops_per_frame']:12d} {r['floor_instr_per_frame']:17.1f}")
    print(f"# per IQ sample: oracle {r['oracle_flops_per_frame'] / (s['Ts'] * s['Nsym']):.1f} flops, floor {r['floor_ops_per_sample']:.1f} lane operations (an fma counts once), "
          f"floor {r['floor_instr_per_frame'] / (s['Ts'] * s['Nsym']):.3f} wave instructions")
    for p in r["phases"]:
        if p["note"]:
            print(f"#   {p['phase']}: {p['note']}")
    if a.executed:
        ex = [float(x) for x in a.executed.split(",")]
        print(f"# {'kernel phase':88s} {'floor':>8s} {'executed':>9s} {'executed/floor':>15s}")
        for (name, idx), e in zip(KERNEL_PHASES, ex):
            f = sum(r["phases"][i]["floor_instr"] for i in idx)
            print(f"  {name:88s} {f:8.1f} {e:9.1f} {e / f:15.2f}")
        print(f"  {'whole frame':88s} {r['floor_instr_per_frame']:8.1f} {sum(ex):9.1f} {sum(ex) / r['floor_instr_per_frame']:15.2f}")


if __name__ == "__main__":
    main()
