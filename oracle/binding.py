"""ctypes binding of the CPU oracle (oracle/build/libpirip_oracle.so).

TEST INFRASTRUCTURE ONLY -- parity unpinned (see oracle/fsk_oracle.h). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under pirip_amd/ may.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libpirip_oracle.so")

IN_CU8_FSKDEMOD, IN_CU8_CSDR, IN_CS16, IN_CF32 = 0, 1, 2, 3


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.oracle_fsk_create_hbr.restype = C.c_void_p
    lib.oracle_fsk_create_hbr.argtypes = [C.c_int] * 7
    lib.oracle_fsk_create_recalled.restype = C.c_void_p
    lib.oracle_fsk_create_recalled.argtypes = [C.c_int] * 7 + [C.c_void_p]
    lib.oracle_fsk_destroy.argtypes = [C.c_void_p]
    lib.oracle_fsk_set_freq_est_limits.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.oracle_fsk_set_freq_est_alg.argtypes = [C.c_void_p, C.c_int]
    lib.oracle_fsk_enable_burst_mode.argtypes = [C.c_void_p]
    lib.oracle_fsk_nin.restype = C.c_uint32
    lib.oracle_fsk_nin.argtypes = [C.c_void_p]
    lib.oracle_fsk_mod_c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_fsk_mod.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_get_test_bits.argtypes = [C.c_void_p, C.c_long, C.c_int]
    lib.oracle_demod_buffer.restype = C.c_long
    lib.oracle_demod_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]

    class PutResult(C.Structure):
        _fields_ = [("packetcnt", C.c_int), ("bitcnt", C.c_long), ("biterr", C.c_long),
                    ("ber", C.c_float), ("passed", C.c_int)]
    lib.oracle_put_test_bits.restype = PutResult
    lib.oracle_put_test_bits.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_int, C.c_float]
    lib.oracle_convert_u8_f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_convert_f_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_firdes_filter_len.restype = C.c_int
    lib.oracle_firdes_filter_len.argtypes = [C.c_float]
    lib.oracle_firdes_lowpass_f_hamming.argtypes = [C.c_void_p, C.c_int, C.c_float]
    lib.oracle_fir_decimate_cc.restype = C.c_int
    lib.oracle_fir_decimate_cc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.oracle_csdr_fir_decimate_stream.restype = C.c_long
    lib.oracle_csdr_fir_decimate_stream.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                                    C.c_int, C.c_float, C.c_int]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleRecalled(C.Structure):
    """struct fsk_oracle_recalled (fsk_oracle.h): the constants held from recall of codec2, as data -- the product's pirip_fsk_recalled."""
    _fields_ = [("hann_denominator_ndft", C.c_int), ("tc", C.c_float), ("est_space_rs", C.c_float), ("nin_threshold", C.c_float),
                ("nin_step_div", C.c_int), ("s16_scale", C.c_float), ("u8d_offset", C.c_float), ("u8d_scale", C.c_float),
                ("ndft_rule", C.c_int), ("sf_power", C.c_int)]


# every field with the alternative value the pin-day drill tries (tests/test_gpu_parity.py, oracle/pin_against_ref.py)
RECALLED_ALTERNATIVES = {"hann_denominator_ndft": 1, "tc": 0.2, "est_space_rs": 1.5, "nin_threshold": 0.3, "nin_step_div": 2, "s16_scale": 1000.0,
                         "u8d_offset": 127.5, "u8d_scale": 127.5, "ndft_rule": 1, "sf_power": 1}


def recalled(**overrides):
    r = OracleRecalled()
    lib().oracle_fsk_recalled_defaults(C.byref(r))
    for k, v in overrides.items():
        if k not in dict(OracleRecalled._fields_):
            raise KeyError(k)
        setattr(r, k, v)
    return r


class OracleFsk:
    """One stream of the oracle demod/mod (struct ORACLE_FSK). recalled=dict(field=value, ...): oracle_fsk_create_recalled."""

    def __init__(self, Fs, Rs, M, P=8, Nsym=50, f1_tx=-1, tone_spacing=100,
                 est_min=None, est_max=None, mask=False, recalled=None):
        self.l = lib()
        rc = globals()["recalled"](**(recalled or {}))
        self.h = self.l.oracle_fsk_create_recalled(Fs, Rs, M, P, Nsym, f1_tx, tone_spacing, C.byref(rc))
        self.Fs, self.Rs, self.M, self.P, self.Nsym = Fs, Rs, M, P, Nsym
        self.Ts = Fs // Rs
        self.nin_step = self.Ts // rc.nin_step_div
        self.N = self.Ts * Nsym
        self.Nbits = Nsym * (1 if M == 2 else 2)
        if est_min is not None:
            self.l.oracle_fsk_set_freq_est_limits(self.h, est_min, est_max)
        if mask:
            self.l.oracle_fsk_set_freq_est_alg(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.oracle_fsk_destroy(self.h)
            self.h = None

    def enable_burst_mode(self):
        self.l.oracle_fsk_enable_burst_mode(self.h)

    def nin(self):
        return int(self.l.oracle_fsk_nin(self.h))

    def clear_estimators(self):
        self.l.oracle_fsk_clear_estimators.argtypes = [C.c_void_p]
        self.l.oracle_fsk_clear_estimators(self.h)

    def snr(self):
        """(smoothed EbNodB = MODEM_STATS.snr_est, EbNodB, v_est) after the last demodulated frame."""
        out = np.zeros(3, dtype=np.float32)
        self.l.oracle_fsk_get_snr.argtypes = [C.c_void_p, C.c_void_p]
        self.l.oracle_fsk_get_snr(self.h, _p(out))
        return out

    def eye(self, normalise=True):
        """MODEM_STATS.rx_eye of the last demodulated frame: array [neyetr, neyesamp]."""
        out = np.zeros((8, 160), dtype=np.float32)
        ntr, nsamp = C.c_int(0), C.c_int(0)
        self.l.oracle_fsk_get_eye.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.l.oracle_fsk_get_eye(self.h, _p(out), C.byref(ntr), C.byref(nsamp), 1 if normalise else 0)
        return out[:ntr.value, :nsamp.value].copy()

    def mod_c(self, bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        nsym = len(bits) // (1 if self.M == 2 else 2)
        out = np.zeros((nsym * self.Ts, 2), dtype=np.float32)
        self.l.oracle_fsk_mod_c(self.h, _p(out), _p(bits), len(bits))
        return out

    def demod(self, buf, fmt, want_filt=True, want_stats=True):
        """buf: np array (uint8 [n,2] / int16 [n,2] / float32 [n,2]). Returns dict."""
        buf = np.ascontiguousarray(buf)
        nsamp = buf.shape[0]
        maxf = nsamp // (self.N - self.nin_step) + 2
        bits = np.zeros((maxf, self.Nbits), dtype=np.uint8)
        filt = np.zeros((maxf, self.M * self.Nsym), dtype=np.float32) if want_filt else None
        st = np.zeros((maxf, 10), dtype=np.float32) if want_stats else None
        consumed = C.c_long(0)
        cond = np.zeros(maxf, dtype=np.float32)
        self.l.oracle_fsk_set_cond_out.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        self.l.oracle_fsk_set_cond_out(self.h, _p(cond), maxf)
        nf = self.l.oracle_demod_buffer(self.h, fmt, _p(buf), nsamp, _p(bits),
                                        _p(filt) if want_filt else None,
                                        _p(st) if want_stats else None, maxf, C.byref(consumed))
        self.l.oracle_fsk_set_cond_out(self.h, None, 0)
        # timing_cond: per frame |t_c| / sum of |terms| of the fine-timing phasor sum (small = the angle is ill-conditioned)
        return {"nframes": int(nf), "consumed": int(consumed.value), "bits": bits[:nf],
                "rx_filt": filt[:nf] if want_filt else None, "stats": st[:nf] if want_stats else None, "timing_cond": cond[:nf]}


def get_test_bits(nbits, framesize=100):
    out = np.zeros(nbits, dtype=np.uint8)
    lib().oracle_get_test_bits(_p(out), nbits, framesize)
    return out


def put_test_bits(bits, framesize=100, valid_thresh=0.1, packet_pass=0, ber_pass=0.0):
    bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
    r = lib().oracle_put_test_bits(_p(bits), len(bits), framesize, valid_thresh, packet_pass, ber_pass)
    return {"packets": r.packetcnt, "bits": r.bitcnt, "errors": r.biterr, "ber": r.ber, "pass": bool(r.passed)}


def quantise_cu8(x_c, amp=32.0):
    """complex float [n,2] (fsk_mod_c output, peak 2.0) -> interleaved u8 IQ [n,2].
    u8 = clamp(round(127 + amp*x)); amp=32 puts a 2.0-peak tone at half scale (build-chosen,
    the reference has no synthetic u8 generator: SURVEY.md 8d cfg 1)."""
    q = np.rint(127.0 + amp * x_c.astype(np.float64))
    return np.clip(q, 0, 255).astype(np.uint8)


# ---- FSK_LDPC (oracle/ldpc_oracle.c) --------------------------------------------------------------------------------
def parse_code_file(path):
    """The code file format of pirip_amd/csrc/fsk_ldpc.hpp, parsed independently of the product for the oracle."""
    hdr, rows = {}, []
    with open(path) as f:
        lines = [ln for ln in f.read().split("\n") if ln and not ln.startswith("#")]
    i = 0
    while i < len(lines):
        key, *rest = lines[i].split()
        i += 1
        if key == "rows":
            nrows = int(rest[0])
            rows = [sorted(int(c) for c in lines[i + r].split()) for r in range(nrows)]
            break
        hdr[key] = rest
    n, k = int(hdr["n"][0]), int(hdr["k"][0])
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col_idx = np.array([c for r in rows for c in r], dtype=np.int32)
    return dict(name=hdr["name"][0], n=n, k=k, max_iter=int(hdr.get("max_iter", ["15"])[0]),
                uw=np.array([int(b) for b in hdr["uw"]], dtype=np.uint8),
                uw_thresh1=int(hdr.get("uw_thresh1", ["5"])[0]), uw_thresh2=int(hdr.get("uw_thresh2", ["6"])[0]),
                bad_uw_thresh=int(hdr.get("bad_uw_thresh", ["1"])[0]), row_ptr=row_ptr, col_idx=col_idx, rows=rows,
                llr_map=hdr.get("llr_map", ["upstream"])[0])


class OracleLdpc:
    """One FSK_LDPC receiver of the oracle: LLR mapping, decoder and the per-call sync state machine."""

    def __init__(self, code, M, Nsym=50, llr_map=None):
        """llr_map: "upstream" (codec2's fsk_rx_filt_to_llrs as recalled -- the product's default) or "rician"; None = what the code file says"""
        self.l = lib()
        self.l.oracle_ldpc_create.restype = C.c_void_p
        self.l.oracle_ldpc_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7
        self.llr_map = llr_map or code.get("llr_map", "upstream")
        assert self.llr_map in ("upstream", "rician")
        self.l.oracle_ldpc_destroy.argtypes = [C.c_void_p]
        self.l.oracle_ldpc_llr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.l.oracle_ldpc_decode.restype = C.c_int
        self.l.oracle_ldpc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        self.l.oracle_ldpc_rx_call.restype = C.c_int
        self.l.oracle_ldpc_rx_call.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.l.oracle_crc16.restype = C.c_uint16
        self.l.oracle_crc16.argtypes = [C.c_void_p, C.c_int]
        self.code, self.M, self.Nsym = code, M, Nsym
        self.Nbits = Nsym * (1 if M == 2 else 2)
        self.h = self.l.oracle_ldpc_create(code["n"], code["k"], _p(code["row_ptr"]), _p(code["col_idx"]), _p(code["uw"]),
                                           code["max_iter"], code["uw_thresh1"], code["uw_thresh2"], code["bad_uw_thresh"], M, Nsym,
                                           1 if self.llr_map == "rician" else 0)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.oracle_ldpc_destroy(self.h)
            self.h = None

    def llr(self, rx_filt_calls):
        r = np.ascontiguousarray(rx_filt_calls, dtype=np.float32).reshape(-1, self.M * self.Nsym)
        out = np.zeros((r.shape[0], self.Nbits), dtype=np.float32)
        for i in range(r.shape[0]):
            self.l.oracle_ldpc_llr(self.h, _p(r[i]), _p(out[i]))
        return out

    def decode(self, llr_cw):
        llr_cw = np.ascontiguousarray(llr_cw, dtype=np.float32).reshape(-1, self.code["n"])
        bits = np.zeros(llr_cw.shape, dtype=np.uint8)
        ip = np.zeros((llr_cw.shape[0], 2), dtype=np.int32)
        for i in range(llr_cw.shape[0]):
            pcc = C.c_int(0)
            ip[i, 0] = self.l.oracle_ldpc_decode(self.h, _p(llr_cw[i]), _p(bits[i]), C.byref(pcc))
            ip[i, 1] = pcc.value
        return bits, ip

    def rx(self, rx_filt_calls):
        """Per-call receiver: returns status [ncalls], payload [ncalls, k/8], info [ncalls, 10]."""
        r = np.ascontiguousarray(rx_filt_calls, dtype=np.float32).reshape(-1, self.M * self.Nsym)
        nb = self.code["k"] // 8
        status = np.zeros(r.shape[0], dtype=np.uint8)
        payload = np.zeros((r.shape[0], nb), dtype=np.uint8)
        info = np.zeros((r.shape[0], 10), dtype=np.int32)
        for i in range(r.shape[0]):
            status[i] = self.l.oracle_ldpc_rx_call(self.h, _p(r[i]), _p(payload[i]), _p(info[i]))
        return status, payload, info

    def crc16(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        return int(self.l.oracle_crc16(_p(data), len(data)))


class IndepLdpc:
    """The SECOND CPU receiver (oracle/ldpc_independent.c): float32 soft bits, serial sums, double-precision sum-product.
    mode 1 = textbook Rician LLRs, mode 2 = codec2's fsk_rx_filt_to_llrs as recalled [UPSTREAM-RECALLED]. What the mirror
    oracle (OracleLdpc) and the GPU are measured against -- never bit for bit, always as decoded payloads and error rates."""
    RICIAN, UPSTREAM_RECALLED, UPSTREAM_RECALLED_PHI0_RANGE = 1, 2, 3

    def __init__(self, code, M, Nsym=50, mode=1):
        self.l = lib()
        self.l.indep_ldpc_create.restype = C.c_void_p
        self.l.indep_ldpc_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7
        self.l.indep_ldpc_destroy.argtypes = [C.c_void_p]
        self.l.indep_ldpc_llr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.l.indep_ldpc_decode.restype = C.c_int
        self.l.indep_ldpc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        self.l.indep_ldpc_rx_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.l.indep_ln_i0.restype = C.c_double
        self.l.indep_ln_i0.argtypes = [C.c_double]
        self.l.indep_logbesseli0_recalled.restype = C.c_float
        self.l.indep_logbesseli0_recalled.argtypes = [C.c_float]
        self.code, self.M, self.Nsym, self.mode = code, M, Nsym, mode
        self.Nbits = Nsym * (1 if M == 2 else 2)
        self.h = self.l.indep_ldpc_create(code["n"], code["k"], _p(code["row_ptr"]), _p(code["col_idx"]), _p(code["uw"]),
                                          code["max_iter"], code["uw_thresh1"], code["uw_thresh2"], code["bad_uw_thresh"], M, Nsym, mode)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.indep_ldpc_destroy(self.h)
            self.h = None

    def llr(self, rx_filt_calls):
        r = np.ascontiguousarray(rx_filt_calls, dtype=np.float32).reshape(-1, self.M * self.Nsym)
        out = np.zeros((r.shape[0], self.Nbits), dtype=np.float32)
        for i in range(r.shape[0]):
            self.l.indep_ldpc_llr(self.h, _p(r[i]), _p(out[i]))
        return out

    def decode(self, llr_cw):
        llr_cw = np.ascontiguousarray(llr_cw, dtype=np.float32).reshape(-1, self.code["n"])
        bits = np.zeros(llr_cw.shape, dtype=np.uint8)
        ip = np.zeros((llr_cw.shape[0], 2), dtype=np.int32)
        for i in range(llr_cw.shape[0]):
            pcc = C.c_int(0)
            ip[i, 0] = self.l.indep_ldpc_decode(self.h, _p(llr_cw[i]), _p(bits[i]), C.byref(pcc))
            ip[i, 1] = pcc.value
        return bits, ip

    def rx(self, rx_filt_calls):
        r = np.ascontiguousarray(rx_filt_calls, dtype=np.float32).reshape(-1, self.M * self.Nsym)
        nb = self.code["k"] // 8
        status = np.zeros(r.shape[0], dtype=np.uint8)
        payload = np.zeros((r.shape[0], nb), dtype=np.uint8)
        info = np.zeros((r.shape[0], 10), dtype=np.int32)
        self.l.indep_ldpc_rx_stream(self.h, _p(r), r.shape[0], _p(status), _p(payload), _p(info))
        return status, payload, info
