"""pirip_amd -- MI355X-native FSK receive path (IQ -> bits) behind pirip's tool/library boundary.

The product is the C-ABI shared library ``pirip_amd/lib/libpirip_hip.so`` (include/pirip_hip.h)
and the CLI tools under ``pirip_amd/bin``. This package is the thin Python binding used by
bench.py and the tests; it never imports the CPU oracle and has no CPU compute path.
"""
from .binding import (  # noqa: F401
    PiripError, FskParams, HipDemod, HipDecim, lib, lib_path, build, device_count, selftest_sqrt, selftest_div,
    IN_CU8_FSKDEMOD, IN_CU8_CSDR, IN_CS16, IN_CF32, STATS_PER_FRAME,
    HipLdpc, STANDIN_CODE, RX_TRIAL_SYNC, RX_SYNC, RX_BITS, RX_BIT_ERRORS, LDPC_INFO_PER_CALL,
)
