"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/pirip_hip.h declares; the host-side plan and the CPU tools (Tx side / measurement
instrument) agree with the oracle; compute entry points refuse to run without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import sigutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pirip_amd", "bin")


def _declared_functions(header="pirip_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src):
        n = m.group(1)
        if n in ("defined", "sizeof") or n.isupper():
            continue
        names.add(n)
    return sorted(names)


def test_library_exports_every_declared_symbol(built_lib):
    names = _declared_functions()
    assert "pirip_hip_demod_batch" in names and "fsk_demod" in names and "fir_decimate_cc" in names
    missing = [n for n in names if not hasattr(built_lib, n)]
    assert not missing, missing


def test_freedv_api_section_is_exported_and_its_tx_side_runs_without_a_gpu(built_lib, tmp_path):
    """include/pirip_hip.h section F: the FreeDV names upstream's rtl_fsk --code and /root/reference/tx/rpitx_fsk.cpp:33-40,164-165,
    319-325,541 bind. The plain-C program of tests/cprog opens FREEDV_MODE_FSK_LDPC by codename, builds a frame with the Tx-side
    helpers the way rpitx_fsk.cpp:75-83,394-395 does (CRC-16/CCITT-FALSE check value asserted inside) and, given no samples, ends
    without ever touching a device; an unknown codename makes freedv_open_advanced return NULL with a note."""
    import pirip_amd
    names = _declared_functions()
    for n in ("freedv_open_advanced", "freedv_nin", "freedv_rawdatacomprx", "freedv_get_rx_status", "freedv_get_bits_per_modem_frame",
              "freedv_set_frames_per_burst", "freedv_close", "freedv_tx_fsk_ldpc_framer", "freedv_gen_crc16", "freedv_pack", "freedv_unpack"):
        assert n in names and hasattr(built_lib, n), n
    exe = str(tmp_path / "rtl_coded")
    libdir = os.path.dirname(pirip_amd.lib_path())
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cprog", "rtl_fsk_coded_like_upstream.c"), "-L", libdir, "-lpirip_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    env = {k: v for k, v in os.environ.items() if k != "PIRIP_CODE_DIR"}
    p = subprocess.run([exe, "standin_256_512_4", "4", "240000", "10000", "500", "60000"], input=b"", capture_output=True, env=env)
    assert p.returncode == 0, p.stderr
    assert b"tx: bits_per_frame 544 data_bits_per_frame 256" in p.stderr and b"calls 0 frames 0" in p.stderr
    p = subprocess.run([exe, "H_256_512_4", "2", "240000", "10000", "500", "25000"], input=b"", capture_output=True, env=env)
    assert p.returncode == 3 and b"no table for code H_256_512_4" in p.stderr


def test_rccl_helper_library_exports_its_header(built_lib):
    """include/pirip_hip_rccl.h (the one gather of the multi-GPU path) is served by libpirip_hip_rccl.so; symbols only,
    nothing is called without a GPU."""
    import pirip_amd
    so = os.path.join(os.path.dirname(pirip_amd.lib_path()), "libpirip_hip_rccl.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = _declared_functions("pirip_hip_rccl.h")
    assert "pirip_hip_gather_bits" in names and "pirip_hip_rccl_init" in names
    assert all(re.search(r"\bT %s$" % n, out, flags=re.M) for n in names), out
    # and the FSK_LDPC code-file framer refuses a broken table instead of guessing
    bad = os.path.join(ROOT, "tests", "golden", "does_not_exist.code")
    p = subprocess.run([os.path.join(BIN, "fsk_ldpc_framer"), "--code", bad, "--testframes", "1", "/dev/zero", "-"], capture_output=True)
    assert p.returncode == 2 and b"cannot open" in p.stderr


def test_no_cpu_fallback_without_device(built_lib):
    import pirip_amd
    if pirip_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pirip_amd.PiripError, match="no usable HIP device"):
        pirip_amd.HipDemod(240000, 10000, 2, P=24)
    with pytest.raises(pirip_amd.PiripError, match="no usable HIP device"):
        pirip_amd.HipDecim(45)
    # the CLI refuses as well (exit code 2, message on stderr), it does not demodulate on the CPU
    p = subprocess.run([os.path.join(BIN, "fsk_demod"), "-d", "-p", "24", "2", "240000", "10000", "-", "-"],
                       input=b"\x80" * 4800, capture_output=True)
    assert p.returncode == 2 and b"no CPU fallback" in p.stderr and p.stdout == b""


def test_bad_configurations_are_rejected_like_codec2_asserts(built_lib):
    import pirip_amd
    for kw in (dict(Fs=240000, Rs=7000, M=2, P=8),      # Fs % Rs
               dict(Fs=240000, Rs=10000, M=2, P=5),      # Ts % P
               dict(Fs=240000, Rs=10000, M=2, P=2),      # P < 4
               dict(Fs=240000, Rs=10000, M=3, P=8)):     # M
        with pytest.raises(pirip_amd.PiripError, match="bad modem configuration"):
            pirip_amd.HipDemod(**kw)


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.run(["ldd", os.path.join(ROOT, "pirip_amd", "lib", "libpirip_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dp, _, fs in os.walk(os.path.join(ROOT, "pirip_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_cli_get_test_bits_and_mod_match_oracle(built_lib, oracle):
    bits = subprocess.run([os.path.join(BIN, "fsk_get_test_bits"), "-", "1000"], capture_output=True).stdout
    assert np.array_equal(np.frombuffer(bits, dtype=np.uint8), oracle.get_test_bits(1000))
    # fsk_mod -c : complex s16, peak = -a amp
    c = sigutil.CFG1
    p = subprocess.run([os.path.join(BIN, "fsk_mod"), "-c", "-a", "30000", "2", "240000", "10000", "10000", "10000", "-", "-"],
                       input=bits, capture_output=True)
    s16 = np.frombuffer(p.stdout, dtype=np.int16).reshape(-1, 2)
    # the CLI modulates Nsym = 50 symbols per fsk_mod_c() call (phase renormalised per call)
    tx = oracle.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], f1_tx=c["f1"], tone_spacing=c["shift"])
    b = np.frombuffer(bits, dtype=np.uint8)
    ref = np.concatenate([tx.mod_c(b[i:i + 50]) for i in range(0, 1000, 50)])
    assert s16.shape[0] == ref.shape[0] == 1000 * 24
    assert np.array_equal(s16, (ref * np.float32(15000.0)).astype(np.int16))


def test_cli_put_test_bits_verdict(built_lib, oracle):
    exe = os.path.join(BIN, "fsk_put_test_bits")
    good = oracle.get_test_bits(10000).tobytes()
    p = subprocess.run([exe, "-q", "-p", "90", "-"], input=good, capture_output=True)
    assert p.returncode == 0 and b"PASS" in p.stderr
    bad = bytearray(good)
    for i in range(0, len(bad), 3):
        bad[i] ^= 1
    p = subprocess.run([exe, "-q", "-p", "90", "-"], input=bytes(bad), capture_output=True)
    assert p.returncode == 1 and b"FAIL" in p.stderr
    # same counts as the oracle's counter
    p = subprocess.run([exe, "-q", "-"], input=good, capture_output=True)
    res = oracle.put_test_bits(np.frombuffer(good, dtype=np.uint8))
    assert f"bits tested {res['bits']:6d}".encode() in p.stderr


def test_exact_input_conversions_used_by_the_kernels():
    """The kernels convert raw samples where they are used, with FMA sequences instead of the table
    / the divide: each must equal the defining expression for EVERY input value (float32 FMA emulated
    exactly with fractions)."""
    from fractions import Fraction

    def rnd(fr):                               # Fraction -> nearest float32, ties to even
        f = np.float32(float(fr))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))

    def fma(a, b, c):
        return rnd(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))

    # csdr convert_u8_f: x/127.5 - 1 in double, rounded to float
    c_hi = np.float32(np.round((1.0 / 127.5) * 2 ** 22) / 2 ** 22)
    c_lo = np.float32(1.0 / 127.5 - float(c_hi))
    for x in range(256):
        want = np.float32(np.float64(np.float32(x)) / (255 / 2.0) - 1.0)
        assert fma(np.float32(x), c_lo, fma(np.float32(x), c_hi, np.float32(-1.0))) == want, x
        # fsk_demod -d: (x - 127)/128
        assert fma(np.float32(x), np.float32(0.0078125), np.float32(-0.9921875)) == np.float32((x - 127) / 128.0)
    # fsk_demod -c: (float)s16 / FDMDV_SCALE as multiply + one Newton step on the residual
    hdr = open(os.path.join(ROOT, "include", "pirip_hip.h")).read()
    scale = np.float32(float(re.search(r"#define\s+PIRIP_FDMDV_SCALE\s+(\d+)", hdr).group(1)))
    assert scale == 750.0
    r = np.float32(1.0) / scale
    vals = list(range(-32768, 32768, 7)) + [-32768, -1, 0, 1, 749, 750, 751, 32767]
    for v in vals:
        q = rnd(Fraction(v) * Fraction(float(r)))
        q2 = fma(fma(-scale, q, np.float32(v)), r, q)
        assert q2 == np.float32(np.float32(v) / scale), v
    # wave-per-stream kernel: x/750 = fma(x, c_lo, x*c_hi) with c_hi = 175/2^17 (8 significant bits, so x*c_hi is
    # exact for every int16) and c_lo the float remainder of 1/750 -- one rounding; every int16 value, vectorised
    c_hi = np.float64(175.0 / 131072.0)
    c_lo = np.float32(1.0 / 750.0 - c_hi)
    assert float(c_lo).hex() == "-0x1.e60f040000000p-20"
    x = np.arange(-32768, 32768).astype(np.float64)
    hi = x * c_hi
    assert np.array_equal(hi.astype(np.float32).astype(np.float64), hi)          # the multiply is exact in float32
    got = (hi + x * np.float64(c_lo)).astype(np.float32)                          # exact in double, one rounding to float
    assert np.array_equal(got, (x.astype(np.float32) / scale).astype(np.float32))


def test_recalled_constants_defaults_are_todays_values_in_product_and_oracle(built_lib):
    """pirip_fsk_recalled (product) and fsk_oracle_recalled (checker) are the same fields in the same order with the same defaults --
    the values that were literals in the kernels and in the restatement until round 6 (no GPU needed: plain data)."""
    import ctypes as C
    import pirip_amd
    from oracle import binding as ob
    assert [f for f, _ in pirip_amd.binding.FskRecalled._fields_] == [f for f, _ in ob.OracleRecalled._fields_]
    assert C.sizeof(pirip_amd.binding.FskRecalled) == C.sizeof(ob.OracleRecalled) == 40
    a, b = pirip_amd.binding.recalled(), ob.recalled()
    want = dict(hann_denominator_ndft=0, tc=np.float32(0.1), est_space_rs=0.75, nin_threshold=0.25, nin_step_div=4, s16_scale=750.0, u8d_offset=127.0,
                u8d_scale=128.0, ndft_rule=0, sf_power=0)
    for k, v in want.items():
        assert getattr(a, k) == getattr(b, k) == v, k
    assert set(ob.RECALLED_ALTERNATIVES) == set(want) and all(ob.RECALLED_ALTERNATIVES[k] != want[k] for k in want)


def test_int16_division_by_the_newton_step_is_the_ieee_quotient_for_any_plausible_scale():
    """The general kernel divides `fsk_demod -c` samples by the plan's s16_scale as q = x * r, q += fma(-scale, q, x) * r (r = 1 / scale
    rounded): exact rational arithmetic shows that this is the correctly rounded float32 quotient for every int16 and for both
    candidate values of FDMDV_SCALE (and two awkward ones), i.e. what the CPU's `(float)x / scale` gives."""
    from fractions import Fraction
    import math

    def rf32(v):
        if v == 0:
            return Fraction(0)
        sgn, a = (1 if v > 0 else -1), abs(v)
        e = math.floor(math.log2(a)) - 23
        while a / Fraction(2) ** e >= 2 ** 24:
            e += 1
        while a / Fraction(2) ** e < 2 ** 23:
            e -= 1
        q = a / Fraction(2) ** e
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return sgn * n * Fraction(2) ** e
    for scale in (750, 1000, 32767):
        sc = Fraction(scale)
        r = rf32(1 / sc)
        for x in range(-32768, 32768, 1 if scale != 32767 else 7):
            X = Fraction(x)
            q0 = rf32(X * r)
            q = rf32(rf32(-sc * q0 + X) * r + q0)
            assert q == rf32(X / sc), (scale, x)
