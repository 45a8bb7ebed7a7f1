"""The decision-level contract at scale: 10^8 bits per case on the Ts = 24 wave instances (2.4 x 10^6 on the
Ts = 240 block instance of `rtl_fsk -r 1000`, README.md:152,184,239), device against an oracle replay of every stream, every differing bit classified (tools/scale_check.py):
  * noise-free: NO bit differs, whatever sample of a symbol the recording starts on (round 5: where a window can hold a single sample,
    P == Ts, the first frame of a created stream is demodulated in the oracle's own operation order -- until then the first decision of a
    recording that starts one sample before a symbol boundary was a rounding coin toss: 34 bits in 10^8);
  * under noise: differences are near-ties of the ORACLE's own decision (margin < 2e-4 of the stream's peak between its two largest
    tone magnitudes, any M), or sit in frames whose fine-timing estimate is ill-conditioned (the two estimates differ by > 5e-5
    symbols), or follow a split of the nin sequence at a timing threshold that both estimates approach to < 5e-5 symbols. Nothing else.
  * the counts of each class stay inside committed bounds (tests/golden/scale_check_bounds.json: what the round-4 kernel gave, with
    headroom for a different noise realisation), so a kernel change that moves them is seen.
What must stay exact under all of it: frame counts, tone estimates, and every bit outside those classes
(/root/reference/test/loopback_rtl_sdr.sh:16 and README.md:105 are the command line these shapes come from)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu
BOUNDS = json.load(open(os.path.join(ROOT, "tests", "golden", "scale_check_bounds.json")))


@pytest.mark.parametrize("case", sorted(BOUNDS["cases"]))
def test_scale_check_decisions_against_the_oracle(oracle, built_lib, case):
    import scale_check
    b = BOUNDS["cases"][case]
    cs = case.split(":")                     # M:P:Eb/N0, or M:P:Eb/N0:streams:Rs:mask spacing:samples for the Ts = 240 block instance
    m, p, e = cs[:3]
    rs, mask, nsamp = (int(cs[4]), int(cs[5]), int(cs[6])) if len(cs) > 4 else (10000, 0, BOUNDS["samples"])
    r = scale_check.run(int(m), int(p), None if e == "none" else float(e), nstreams=b["streams"], nsamp=nsamp, rs=rs, mask=mask)
    print({k: v for k, v in r.items() if k not in ("detail", "worst", "probe", "timing_splits")})
    assert r["bits"] > 0.9 * b["streams"] * (nsamp // (240000 // rs * 50)) * 50 * (1 if m == "2" else 2)
    assert r["frame_count_mismatch"] == 0 and r["fest_mismatch_streams"] == 0
    assert r["outside"] == 0, "a bit differs from the oracle's and is neither a near-tie, nor in an ill-conditioned timing frame"
    assert r["unexplained_splits"] == 0, "a stream's nin sequence parts from the oracle's away from a timing threshold"
    if e == "none":
        assert r["inside"] == 0 and r["first"] == 0, "noise-free input: every bit is the oracle's"
        assert r["first_diffs_on_zero_offset_streams"] == 0 and r["illcond"] == 0 and r["nin_mismatch_streams"] == 0
        assert r["max_filt_err"] < 1e-4
    for key in ("inside", "illcond", "nin_mismatch_streams", "frames_over_1e4", "frames_over_1e3"):
        assert r[key] <= b[key], (key, r[key], b[key])


@pytest.mark.parametrize("case", ["2:24:3", "4:8:5"])
def test_exact_kernel_has_no_differing_word_at_scale(oracle, built_lib, case, monkeypatch):
    """PIRIP_KERNEL=exact (fsk_demod_exact_kernel: the oracle's operation order on the device, every frame) at the NOISIEST cases of the
    table above: 2.5 x 10^7 / 5 x 10^7 bits, none differs, no nin sequence splits, and the largest soft-magnitude error is exactly 0 --
    so every entry of the fast kernels' classes (near-tie, ill-conditioned timing frame, split) is float32 summation order and nothing
    else (profiles/r05_n_scale_check_exact_kernel.txt holds the 2048-stream table: 0 of 9.2 x 10^8)."""
    import scale_check
    monkeypatch.setenv("PIRIP_KERNEL", "exact")
    m, p, e = case.split(":")
    r = scale_check.run(int(m), int(p), float(e), nstreams=512, nsamp=BOUNDS["samples"])
    print({k: v for k, v in r.items() if k not in ("detail", "worst", "probe", "timing_splits")})
    assert "exact" in r["kernel"]
    assert r["bits"] > 0.9 * 512 * (BOUNDS["samples"] // 1200) * 50 * (1 if m == "2" else 2)
    for key in ("inside", "illcond", "outside", "first", "nin_mismatch_streams", "unexplained_splits", "frame_count_mismatch",
                "fest_mismatch_streams", "frames_over_1e4", "frames_over_1e3"):
        assert r[key] == 0, (key, r[key])
    assert r["max_filt_err"] == 0.0
