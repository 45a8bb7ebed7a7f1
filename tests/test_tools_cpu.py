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
