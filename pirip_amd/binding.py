"""ctypes binding of libpirip_hip.so (include/pirip_hip.h, sections A and B).

Device buffers are passed as raw device pointers (ints): with PyTorch, ``tensor.data_ptr()``
and ``torch.cuda.current_stream().cuda_stream``. Nothing here computes on the CPU; if the
library is missing or no HIP device is usable the calls raise PiripError.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIRIP_HIP_LIB: measurement builds only (e.g. the -DPIRIP_WAVE_TIMING library tools/phase_split.py reads its cycle split from)
_LIB = os.environ.get("PIRIP_HIP_LIB") or os.path.join(_HERE, "lib", "libpirip_hip.so")

IN_CU8_FSKDEMOD, IN_CU8_CSDR, IN_CS16, IN_CF32 = 0, 1, 2, 3
STATS_PER_FRAME = 10


class PiripError(RuntimeError):
    pass


class FskParams(C.Structure):
    _fields_ = [("Fs", C.c_int), ("Rs", C.c_int), ("M", C.c_int), ("P", C.c_int), ("Nsym", C.c_int),
                ("est_min", C.c_int), ("est_max", C.c_int), ("freq_est_type", C.c_int),
                ("tone_spacing", C.c_int), ("in_format", C.c_int)]


class FskInfo(C.Structure):
    _fields_ = [("Ts", C.c_int), ("N", C.c_int), ("Nmem", C.c_int), ("Ndft", C.c_int), ("Nbits", C.c_int),
                ("nin_max", C.c_int), ("nstreams", C.c_int), ("bytes_per_sample", C.c_int)]


class CaptureReport(C.Structure):
    _fields_ = [("segments", C.c_int32), ("segment_frames", C.c_int32), ("passes", C.c_int32), ("segments_rerun", C.c_int32),
                ("frames_demodulated", C.c_int64)]


class LdpcInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n", "k", "bits_per_frame", "data_bytes", "nbits_per_call", "max_iter", "nstreams")] + \
               [("name", C.c_char * 64)]


RX_TRIAL_SYNC, RX_SYNC, RX_BITS, RX_BIT_ERRORS = 1, 2, 4, 8
LDPC_INFO_PER_CALL = 10
STANDIN_CODE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "standin_256_512_4.code")


def lib_path():
    return _LIB


def build(verbose=False):
    """Compile libpirip_hip.so + tools in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise PiripError(f"{_LIB} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no fallback implementation)")
    # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so). Two HIP runtimes in
    # one process do not share the device, so when torch is importable load it FIRST: the
    # dynamic loader then satisfies libpirip_hip.so's libamdhip64 dependency from the copy that
    # is already mapped, and device pointers / streams from torch are valid in our calls.
    if os.environ.get("PIRIP_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_LIB)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    L.pirip_hip_version.restype = C.c_char_p
    L.pirip_hip_strerror.restype = C.c_char_p
    L.pirip_hip_strerror.argtypes = [i32]
    L.pirip_hip_device_count.restype = i32
    L.pirip_hip_create.argtypes = [C.POINTER(FskParams), i32, i32, C.POINTER(vp)]
    L.pirip_hip_destroy.argtypes = [vp]
    L.pirip_hip_get_info.argtypes = [vp, C.POINTER(FskInfo)]
    L.pirip_hip_reset.argtypes = [vp, vp]
    L.pirip_hip_demod_batch.argtypes = [vp, vp, sz, i64, vp, sz, vp, sz, vp, sz, vp, vp, i64, vp]
    L.pirip_hip_demod_capture.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(CaptureReport), vp]
    L.pirip_hip_demod_host.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.pirip_hip_nin0.argtypes = [vp]
    L.pirip_hip_get_Sf.argtypes = [vp, i32, vp]
    L.pirip_hip_get_scalars.argtypes = [vp, i32, vp]
    L.pirip_hip_set_burst_mode.argtypes = [vp, i32]
    L.pirip_hip_set_bit_packing.argtypes = [vp, i32]
    L.pirip_hip_set_estimator_band_only.argtypes = [vp, i32]
    L.pirip_hip_set_freq_est_limits.argtypes = [vp, i32, i32]
    L.pirip_hip_decim_create.argtypes = [i32, C.c_float, i32, i32, C.POINTER(vp)]
    L.pirip_hip_decim_destroy.argtypes = [vp]
    L.pirip_hip_decim_taps.argtypes = [vp, vp, C.POINTER(i32)]
    L.pirip_hip_decim_nout.restype = i64
    L.pirip_hip_decim_nout.argtypes = [vp, i64]
    L.pirip_hip_decim_batch.argtypes = [vp, vp, sz, i64, vp, sz, i32, vp]
    L.pirip_hip_synth_cu8.argtypes = [i32, i32, i32, i32, vp, i32, vp, vp, sz, i64, vp, sz, i64,
                                      C.c_float, C.c_float, C.c_uint64, vp]
    L.pirip_hip_ldpc_create.argtypes = [C.c_char_p, i32, i32, i32, i32, C.POINTER(vp)]
    L.pirip_hip_ldpc_destroy.argtypes = [vp]
    L.pirip_hip_ldpc_get_info.argtypes = [vp, C.POINTER(LdpcInfo)]
    L.pirip_hip_ldpc_reset.argtypes = [vp, vp]
    L.pirip_hip_ldpc_rx_batch.argtypes = [vp, vp, sz, vp, i32, vp, vp, vp, vp]
    L.pirip_hip_ldpc_rx_host.argtypes = [vp, vp, i32, vp, vp, vp]
    L.pirip_hip_fsk_ldpc_rx_batch.argtypes = [vp, vp, vp, sz, C.c_int64, vp, vp, vp, vp, sz, vp, vp, C.c_int64, vp]
    L.pirip_hip_fsk_ldpc_last_path.argtypes = [vp]
    L.pirip_hip_ldpc_llr.argtypes = [vp, vp, i32, vp, vp]
    L.pirip_hip_ldpc_decode_llr.argtypes = [vp, vp, i32, vp, vp, vp]
    _lib = L
    return L


def device_count():
    return int(lib().pirip_hip_device_count())


def selftest_sqrt():
    """Mismatches of the wave kernel's correctly rounded square roots against (float)sqrt((double)x) over x = 0 and every
    float in [2^-96, FLT_MAX], counted on the device (pirip_hip_selftest_sqrt): (v_sqrt variant << 32) | rsq variant."""
    L = lib()
    m = C.c_uint64(0)
    L.pirip_hip_selftest_sqrt.argtypes = [C.POINTER(C.c_uint64)]
    _chk(L.pirip_hip_selftest_sqrt(C.byref(m)), "pirip_hip_selftest_sqrt")
    return int(m.value)


def selftest_div():
    """Mismatches of the fused hand-over's x / 3 and x / 50 (pirip_hip_selftest_div) against the device's IEEE quotient over x = 0 and
    every float in [2^-125, FLT_MAX]: (count for / 50 << 32) | count for / 3."""
    L = lib()
    m = C.c_uint64(0)
    L.pirip_hip_selftest_div.argtypes = [C.POINTER(C.c_uint64)]
    _chk(L.pirip_hip_selftest_div(C.byref(m)), "pirip_hip_selftest_div")
    return int(m.value)


def _chk(rc, what):
    if rc != 0:
        raise PiripError(f"{what}: {lib().pirip_hip_strerror(rc).decode()} ({rc})")


class FskRecalled(C.Structure):
    """pirip_fsk_recalled: the constants held from recall of codec2, as data (include/pirip_hip.h)."""
    _fields_ = [("hann_denominator_ndft", C.c_int), ("tc", C.c_float), ("est_space_rs", C.c_float), ("nin_threshold", C.c_float),
                ("nin_step_div", C.c_int), ("s16_scale", C.c_float), ("u8d_offset", C.c_float), ("u8d_scale", C.c_float),
                ("ndft_rule", C.c_int), ("sf_power", C.c_int)]


def recalled(**overrides):
    """pirip_hip_recalled_defaults() with the named fields replaced."""
    r = FskRecalled()
    lib().pirip_hip_recalled_defaults(C.byref(r))
    for k, v in overrides.items():
        if k not in dict(FskRecalled._fields_):
            raise KeyError(k)
        setattr(r, k, v)
    return r


class HipDemod:
    """nstreams device-resident demodulators (pirip_hip_create; recalled=dict(field=value, ...): pirip_hip_create_recalled)."""

    def __init__(self, Fs, Rs, M, P=8, Nsym=50, est_min=0, est_max=0, mask=0, in_format=IN_CU8_FSKDEMOD,
                 nstreams=1, device=-1, recalled=None):
        self.L = lib()
        self.params = FskParams(Fs, Rs, M, P, Nsym, est_min, est_max, 1 if mask else 0, mask if mask else 100, in_format)
        h = C.c_void_p()
        if recalled is None:
            _chk(self.L.pirip_hip_create(C.byref(self.params), nstreams, device, C.byref(h)), "pirip_hip_create")
        else:
            rc = globals()["recalled"](**recalled)
            _chk(self.L.pirip_hip_create_recalled(C.byref(self.params), C.byref(rc), nstreams, device, C.byref(h)), "pirip_hip_create_recalled")
        self.h = h
        self.info = FskInfo()
        _chk(self.L.pirip_hip_get_info(self.h, C.byref(self.info)), "pirip_hip_get_info")
        self.nstreams = nstreams
        self.M, self.Nsym, self.Nbits, self.N = M, Nsym, self.info.Nbits, self.info.N

    def close(self):
        if getattr(self, "h", None):
            self.L.pirip_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def kernel(self):
        """'wave' / 'block' (a specialised instance), 'general', or 'exact' (PIRIP_KERNEL=exact: every frame in the CPU algorithm's own
        operation order, the on-device cross-check) -- pirip_hip_get_kernel."""
        self.L.pirip_hip_get_kernel.argtypes = [C.c_void_p]
        k = self.L.pirip_hip_get_kernel(self.h)
        return "wave" if k == 2 else "block" if k == 3 else "exact" if k == 4 else "general"

    def kernel_name(self):
        """pirip_hip_get_kernel_name: the instance's template arguments / the general kernel's run-time shape."""
        buf = C.create_string_buffer(256)
        self.L.pirip_hip_get_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        _chk(self.L.pirip_hip_get_kernel_name(self.h, buf, 256), "pirip_hip_get_kernel_name")
        return buf.value.decode()

    def reset(self, stream=0):
        _chk(self.L.pirip_hip_reset(self.h, stream), "pirip_hip_reset")

    def max_frames_for(self, nsamp):
        return nsamp // (2 * self.N - self.info.nin_max) + 2          # (the shortest frame: N - the nin step)

    def demod_batch(self, d_in, in_stride, nsamp, d_bits=0, bits_stride=0, d_filt=0, filt_stride=0,
                    d_stats=0, stats_stride=0, d_nframes=0, d_consumed=0, max_frames=None, stream=0):
        """Raw device pointers (ints); enqueues on `stream`, does not synchronise."""
        if max_frames is None:
            max_frames = self.max_frames_for(nsamp)
        _chk(self.L.pirip_hip_demod_batch(self.h, d_in, in_stride, nsamp, d_bits, bits_stride, d_filt, filt_stride,
                                          d_stats, stats_stride, d_nframes, d_consumed, max_frames, stream),
             "pirip_hip_demod_batch")

    def demod_capture(self, d_in, nsamp, d_bits, d_filt=0, d_stats=0, max_frames=None, stream=0):
        """One long capture on the handle's stream slots (pirip_hip_demod_capture): raw device pointers; synchronises.
        Returns (nframes, consumed, report dict)."""
        if max_frames is None:
            max_frames = self.max_frames_for(nsamp)
        nf, cons, rep = C.c_int64(0), C.c_int64(0), CaptureReport()
        _chk(self.L.pirip_hip_demod_capture(self.h, d_in, nsamp, d_bits, d_filt, d_stats, max_frames, C.byref(nf), C.byref(cons),
                                            C.byref(rep), stream), "pirip_hip_demod_capture")
        return nf.value, cons.value, {k: getattr(rep, k) for k, _ in CaptureReport._fields_}

    def demod_host(self, buf, want_filt=True):
        """numpy buffer [n, 2] in the configured format -> dict (stream 0; uploads/downloads)."""
        import numpy as np
        buf = np.ascontiguousarray(buf)
        nsamp = buf.shape[0]
        maxf = self.max_frames_for(nsamp)
        fb = (self.Nbits + 7) // 8 if getattr(self, "packed", False) else self.Nbits
        bits = np.zeros((maxf, fb), dtype=np.uint8)
        filt = np.zeros((maxf, self.M * self.Nsym), dtype=np.float32)
        st = np.zeros((maxf, STATS_PER_FRAME), dtype=np.float32)
        nf, cons = C.c_int64(0), C.c_int64(0)
        _chk(self.L.pirip_hip_demod_host(self.h, buf.ctypes.data, nsamp, bits.ctypes.data,
                                         filt.ctypes.data if want_filt else None, st.ctypes.data, maxf,
                                         C.byref(nf), C.byref(cons)), "pirip_hip_demod_host")
        n = nf.value
        return {"nframes": n, "consumed": cons.value, "bits": bits[:n], "rx_filt": filt[:n] if want_filt else None,
                "stats": st[:n]}

    def set_bit_packing(self, packed=True):
        """d_bits becomes ceil(Nbits/8) bytes per frame, MSB first (strides in packed bytes)."""
        _chk(self.L.pirip_hip_set_bit_packing(self.h, 1 if packed else 0), "pirip_hip_set_bit_packing")
        self.packed = bool(packed)

    def set_exact_first_frame(self, enable=True):
        """pirip_hip_set_exact_first_frame: the prologue that demodulates a created stream's first frame in the CPU restatement's own
        operation order where P == Ts (default on); off: frame 0 comes from the handle's kernel like every other frame (A/B runs)."""
        self.L.pirip_hip_set_exact_first_frame.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.pirip_hip_set_exact_first_frame(self.h, 1 if enable else 0), "pirip_hip_set_exact_first_frame")

    def set_estimator_band_only(self, enable=True):
        """Opt-in: Sf is maintained only for the FFT bins the peak search can read (include/pirip_hip.h); outputs unchanged."""
        _chk(self.L.pirip_hip_set_estimator_band_only(self.h, 1 if enable else 0), "pirip_hip_set_estimator_band_only")

    def set_freq_est_limits(self, est_min, est_max):
        """fsk_set_freq_est_limits() on the live handle; returns the status code (PIRIP_OK = 0) instead of raising."""
        return int(self.L.pirip_hip_set_freq_est_limits(self.h, int(est_min), int(est_max)))

    def set_burst_mode(self, enable=True):
        _chk(self.L.pirip_hip_set_burst_mode(self.h, 1 if enable else 0), "pirip_hip_set_burst_mode")

    def enable_eye(self, enable=True):
        """MODEM_STATS.rx_eye: keep the eye traces of each stream's latest frame (moves the handle to the general kernel, resets state)."""
        self.L.pirip_hip_enable_eye.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.pirip_hip_enable_eye(self.h, 1 if enable else 0), "pirip_hip_enable_eye")

    def eye(self, s=0, normalise=True):
        import numpy as np
        out = np.zeros((8, 160), dtype=np.float32)
        ntr, npt = C.c_int(0), C.c_int(0)
        self.L.pirip_hip_get_eye.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.pirip_hip_get_eye(self.h, s, 1 if normalise else 0, out.ctypes.data, C.byref(ntr), C.byref(npt)), "pirip_hip_get_eye")
        return out[:ntr.value, :npt.value].copy()

    def get_Sf(self, s=0):
        import numpy as np
        out = np.zeros(self.info.Ndft, dtype=np.float32)
        _chk(self.L.pirip_hip_get_Sf(self.h, s, out.ctypes.data), "pirip_hip_get_Sf")
        return out


class HipDecim:
    """csdr convert_u8_f | fir_decimate_cc D | [convert_f_s16] as one device stage."""

    def __init__(self, D, transition_bw=0.05, out_s16=True, device=-1):
        self.L = lib()
        h = C.c_void_p()
        _chk(self.L.pirip_hip_decim_create(D, transition_bw, 1 if out_s16 else 0, device, C.byref(h)),
             "pirip_hip_decim_create")
        self.h, self.D, self.out_s16 = h, D, out_s16

    def close(self):
        if getattr(self, "h", None):
            self.L.pirip_hip_decim_destroy(self.h)
            self.h = None

    __del__ = close

    def taps(self):
        import numpy as np
        n = C.c_int(0)
        self.L.pirip_hip_decim_taps(self.h, None, C.byref(n))
        t = np.zeros(n.value, dtype=np.float32)
        self.L.pirip_hip_decim_taps(self.h, t.ctypes.data, C.byref(n))
        return t

    def nout(self, n_in):
        return int(self.L.pirip_hip_decim_nout(self.h, n_in))

    def set_arith(self, mode):
        """0 exact (default: the scalar csdr loop bit for bit), 1 fused accumulate, 2 affine map pulled out of the sum (opt-in measurements)"""
        _chk(self.L.pirip_hip_decim_set_arith(self.h, int(mode)), "pirip_hip_decim_set_arith")

    def batch(self, d_in, in_stride, n_in, d_out, out_stride, nstreams, stream=0):
        _chk(self.L.pirip_hip_decim_batch(self.h, d_in, in_stride, n_in, d_out, out_stride, nstreams, stream),
             "pirip_hip_decim_batch")


def synth_cu8(Fs, Rs, M, f1_hz, tone_spacing, d_bits, bits_stride, nsym, d_out, out_stride, nsamp,
              amp=32.0, sigma=0.0, seed=1, skip=None, stream=0):
    """Device-side fsk_mod -c | u8 quantiser [| AWGN] for len(f1_hz) streams (include/pirip_hip.h B2).
    d_bits / d_out are device pointers; f1_hz / skip are host integer sequences."""
    import numpy as np
    f1 = np.ascontiguousarray(f1_hz, dtype=np.int32)
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.int32)
    assert sk is None or sk.size == f1.size
    _chk(lib().pirip_hip_synth_cu8(Fs, Rs, M, int(f1.size), f1.ctypes.data, tone_spacing,
                                   None if sk is None else sk.ctypes.data, d_bits, bits_stride, nsym,
                                   d_out, out_stride, nsamp, amp, sigma, seed, stream), "pirip_hip_synth_cu8")


class ChainGroup(C.Structure):
    """struct pirip_chain_group (include/pirip_hip.h)"""
    _fields_ = [("dem", C.c_void_p), ("ldpc", C.c_void_p), ("d_in", C.c_void_p), ("d_status", C.c_void_p), ("d_payload", C.c_void_p),
                ("d_info", C.c_void_p), ("d_stats", C.c_void_p), ("d_nframes", C.c_void_p), ("d_consumed", C.c_void_p)]


class HipLdpc:
    """nstreams FSK_LDPC receivers (include/pirip_hip.h section E): soft decisions -> status / payload records."""

    def __init__(self, code_path, M, Nsym=50, nstreams=1, device=-1):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.pirip_hip_ldpc_create(code_path.encode(), M, Nsym, nstreams, device, C.byref(self.h)), "pirip_hip_ldpc_create")
        self.info = LdpcInfo()
        _chk(self.L.pirip_hip_ldpc_get_info(self.h, C.byref(self.info)), "pirip_hip_ldpc_get_info")
        self.M, self.Nsym, self.nstreams = M, Nsym, nstreams
        self.n, self.k, self.Nbits, self.data_bytes = self.info.n, self.info.k, self.info.nbits_per_call, self.info.data_bytes

    def close(self):
        if self.h:
            self.L.pirip_hip_ldpc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=0):
        _chk(self.L.pirip_hip_ldpc_reset(self.h, stream), "pirip_hip_ldpc_reset")

    def rx_batch(self, d_rx_filt, filt_stride, d_ncalls, ncalls, d_status, d_payload, d_info, stream=0):
        _chk(self.L.pirip_hip_ldpc_rx_batch(self.h, d_rx_filt, filt_stride, d_ncalls, ncalls, d_status, d_payload, d_info, stream),
             "pirip_hip_ldpc_rx_batch")

    def chain_batch(self, dem, d_in, in_stride, nsamp, d_status, d_payload, d_info, d_nframes, d_consumed, max_frames,
                    d_stats=0, stats_stride=0, stream=0):
        """pirip_hip_fsk_ldpc_rx_batch: IQ of every stream of HipDemod `dem` -> records of this receiver's streams (device pointers)."""
        _chk(self.L.pirip_hip_fsk_ldpc_rx_batch(dem.h, self.h, d_in, in_stride, nsamp, d_status, d_payload, d_info, d_stats, stats_stride,
                                                d_nframes, d_consumed, max_frames, stream), "pirip_hip_fsk_ldpc_rx_batch")

    @staticmethod
    def chain_batch_groups(groups, in_stride, nsamp, max_frames, stats_stride=0, stream=0):
        """pirip_hip_fsk_ldpc_rx_batch_groups: groups = [(HipLdpc, HipDemod, d_in, d_status, d_payload, d_info, d_nframes, d_consumed[, d_stats]), ...]"""
        arr = (ChainGroup * len(groups))()
        for i, g in enumerate(groups):
            ld, dem, d_in, d_status, d_payload, d_info, d_nframes, d_consumed = g[:8]
            arr[i] = ChainGroup(dem.h, ld.h, d_in, d_status, d_payload, d_info, g[8] if len(g) > 8 else None, d_nframes, d_consumed)
        L = lib()
        L.pirip_hip_fsk_ldpc_rx_batch_groups.argtypes = [C.POINTER(ChainGroup), C.c_int, C.c_size_t, C.c_int64, C.c_size_t, C.c_int64, C.c_void_p]
        _chk(L.pirip_hip_fsk_ldpc_rx_batch_groups(arr, len(groups), in_stride, nsamp, stats_stride, max_frames, stream), "pirip_hip_fsk_ldpc_rx_batch_groups")

    def last_path_fused(self):
        return self.L.pirip_hip_fsk_ldpc_last_path(self.h) == 1

    def rx_host(self, rx_filt_calls):
        import numpy as np
        r = np.ascontiguousarray(rx_filt_calls, dtype=np.float32).reshape(-1, self.M * self.Nsym)
        n = r.shape[0]
        status = np.zeros(n, dtype=np.uint8)
        payload = np.zeros((n, self.data_bytes), dtype=np.uint8)
        info = np.zeros((n, LDPC_INFO_PER_CALL), dtype=np.int32)
        _chk(self.L.pirip_hip_ldpc_rx_host(self.h, r.ctypes.data, n, status.ctypes.data, payload.ctypes.data, info.ctypes.data),
             "pirip_hip_ldpc_rx_host")
        return status, payload, info
