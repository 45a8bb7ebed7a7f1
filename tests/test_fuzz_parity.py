"""A fixed slice of tools/fuzz_parity.py in the GPU suite: 400 random configurations / formats / noise levels / clock offsets / call
splits (half of them shapes the reference's command lines use, i.e. the wave and block instances), device against oracle with the
rules of DESIGN.md 5. The long runs (10^5 draws) are in profiles/r04_fuzz_parity.txt."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def test_four_hundred_random_draws_have_nothing_unexplained(oracle, built_lib):
    import fuzz_parity
    import pirip_amd
    import sigutil
    import test_gpu_parity as cmp
    counts, kernels, fails = {}, set(), []
    for seed in range(300000, 300400):
        cfg = fuzz_parity.draw(seed)
        res, msg, kern = fuzz_parity.run_one(cfg, oracle, pirip_amd, sigutil, cmp)
        counts[res] = counts.get(res, 0) + 1
        kernels.add(kern)
        if res == "FAIL":
            fails.append((seed, cfg, msg))
    print(counts)
    assert not fails, fails[:3]
    assert counts.get("exact", 0) > 300 and {"wave", "block", "general"} <= kernels


def test_two_thousand_random_decimator_draws_are_bit_exact(oracle, built_lib):
    """csdr front end: random decimation, tap count, stream count, length, byte alignment and stride; f32 and s16 outputs bit for bit."""
    import fuzz_parity
    import pirip_amd
    fails = [(seed, r[1]) for seed in range(500000, 502000) for r in [fuzz_parity.decim_one(seed, oracle, pirip_amd)] if r[0] != "exact"]
    assert not fails, fails[:3]


def test_one_hundred_random_batches_have_nothing_unexplained(oracle, built_lib):
    """pirip_hip_demod_batch: 1 .. 9 streams per draw at a random stride, a frame cap in some: every stream against its own oracle replay."""
    import fuzz_parity
    import pirip_amd
    import sigutil
    import test_gpu_parity as cmp
    fails, kernels = [], set()
    for seed in range(800000, 800100):
        res, msg, kern = fuzz_parity.batch_one(fuzz_parity.draw(seed), oracle, pirip_amd, sigutil, cmp)
        kernels.add(kern)
        if res == "FAIL":
            fails.append((seed, msg))
    assert not fails, fails[:3]
    assert {"wave", "general"} <= kernels


def test_sixty_random_fsk_ldpc_receptions_equal_the_mirror_oracle(oracle, built_lib):
    """FSK_LDPC receive: random M / P / burst pattern / Eb/N0 2 .. 10 dB / call chunking: status, payload and info records bit for bit."""
    import fuzz_parity
    import pirip_amd
    import sigutil
    fails = [(seed, r[1]) for seed in range(950000, 950060) for r in [fuzz_parity.ldpc_one(seed, oracle, pirip_amd, sigutil)] if r[0] != "exact"]
    assert not fails, fails[:3]


def test_forty_random_iq_to_records_chains_equal_demodulator_plus_oracle_receiver(oracle, built_lib):
    """pirip_hip_fsk_ldpc_rx_batch over 1 .. 6 streams: fused and unfused routes, records equal to unfused demodulator + oracle receiver."""
    import fuzz_parity
    import pirip_amd
    import sigutil
    res = [fuzz_parity.chain_one(seed, oracle, pirip_amd, sigutil) for seed in range(1200000, 1200040)]
    assert all(r[0] == "exact" for r in res), [r for r in res if r[0] != "exact"][:3]
    assert {r[1] for r in res} == {"fused", "unfused"}


def test_sixty_random_recordings_through_the_capture_route_equal_the_read_loop(oracle, built_lib):
    """pirip_hip_demod_capture against the same library's read loop, bit for bit (arrays, counts, state): random shape / noise / clock offset /
    work slots / segment length / pieces."""
    import fuzz_parity
    import pirip_amd
    import sigutil
    res = [fuzz_parity.capture_one(seed, oracle, pirip_amd, sigutil) for seed in range(6200000, 6200060)]
    assert all(r[0] == "exact" for r in res), [r for r in res if r[0] != "exact"][:3]
