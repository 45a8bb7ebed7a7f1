"""CPU checks of the measurement tools' own arithmetic (nothing here touches a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_fft_operation_counts_follow_kiss_fft_factorisation():
    import valu_floor as vf
    assert vf.fft_factors(256) == [4, 4, 4, 4] and vf.fft_factors(512) == [4, 4, 4, 4, 2] and vf.fft_factors(128) == [4, 4, 4, 2]
    # Ndft = 256: 4 stages x 64 radix-4 butterflies; the k = 0 butterfly of every group has twiddle (1,-0): 64 + 16 + 4 + 1 of them
    cadd, need, execd = vf.fft_ops(256)
    assert cadd == 4 * 64 * 8 and execd == 4 * 64 * 3 and need == (256 - 85) * 3
    # Ndft = 512: the radix-2 leaf stage (m = 1) is all trivial twiddles
    cadd, need, execd = vf.fft_ops(512)
    assert cadd == 256 * 2 + 4 * 128 * 8 and execd == 256 + 4 * 128 * 3
    assert need == (128 - 64) * 3 + (128 - 16) * 3 + (128 - 4) * 3 + (128 - 1) * 3


def test_headline_floor_matches_the_figure_quoted_in_design_and_bench():
    import valu_floor as vf
    r = vf.floor(2, 24, 24, 50, 256, "u8")
    assert r["shape"]["nfft"] == 8 and r["shape"]["nint"] == 1224
    assert abs(r["floor_instr_per_frame"] - 1174.7) < 0.1
    # the oracle's own flop count per sample is dominated by the O(P)-redundant window sums the floor does not need
    assert 200 < r["oracle_flops_per_frame"] / 1200 < 240
    # 4-FSK needs more (two more oscillators) and never less than the estimator alone
    r4 = vf.floor(4, 24, 8, 50, 256, "u8")
    est = sum(p["floor_instr"] for p in r4["phases"][:4])
    assert r4["floor_instr_per_frame"] > r["floor_instr_per_frame"] > est > 700


def test_block_shape_floor_matches_the_figure_quoted_in_design():
    """Ts = 240 / P = 15 / Ndft = 4096 (`rtl_fsk -r 1000` at 240 kS/s): six radix-4 stages, four FFTs per 12 000-sample frame."""
    import valu_floor as vf
    assert vf.fft_factors(4096) == [4, 4, 4, 4, 4, 4]
    r = vf.floor(2, 240, 15, 50, 4096, "u8csdr")
    assert r["shape"]["nfft"] == 4 and r["shape"]["nint"] == 765
    assert abs(r["floor_instr_per_frame"] - 10860.3) < 0.5
    assert 0.89 < r["floor_instr_per_frame"] / 12000 < 0.92          # wave instructions per sample: a little below the headline shape's 0.98


def test_band_only_floor_counts_only_what_feeds_the_band():
    """the opt-in band-only estimator: the pruned butterfly count equals the full one when every bin is wanted, drops the
    dead outputs of the last two stages for bins 0..31, and the |X| / sqrt / IIR work scales with the band"""
    import valu_floor as vf
    assert vf.fft_ops_pruned(256, range(256)) == vf.fft_ops(256)[:2]
    assert vf.fft_ops_pruned(512, range(512)) == vf.fft_ops(512)[:2] and vf.fft_ops_pruned(4096, range(4096)) == vf.fft_ops(4096)[:2]
    cadd, cmul = vf.fft_ops_pruned(256, range(32))
    # stages 1, 2 whole (2 x 64 x 8 adds); stage 3: 64 butterflies with outputs {0, 1} (2+1 + 2+1 adds); stage 4: butterflies 0..31, output 0 (3 adds)
    assert cadd == 2 * 64 * 8 + 64 * 6 + 32 * 3 and cmul == (64 - 64) * 3 + (64 - 16) * 3 + (64 - 4) * 3 + (32 - 1) * 3
    full, band = vf.floor(2, 24, 24, 50, 256, "u8"), vf.floor(2, 24, 24, 50, 256, "u8", band_bins=32)
    assert abs(band["floor_instr_per_frame"] - 860.7) < 0.1 and band["floor_instr_per_frame"] < full["floor_instr_per_frame"]
    assert [p["floor_instr"] for p in band["phases"][5:]] == [p["floor_instr"] for p in full["phases"][5:]]     # the correlator side is untouched
